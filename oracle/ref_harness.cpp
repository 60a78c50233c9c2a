// oracle/_ref driver: runs the REFERENCE's own, unmodified CPU sources for this path.  TEST INFRASTRUCTURE ONLY.
//
// Built by oracle/build_ref.sh into oracle/_ref/ref_lmp together with these files compiled where they lie under
// /root/reference/src (nothing is copied into the repository, nothing of the reference is edited or replaced):
//   random_mars.cpp random_park.cpp pair_dpd.cpp pair.cpp fix_nve.cpp fix.cpp neighbor.cpp neigh_half_bin.cpp
//   neigh_stencil.cpp neigh_list.cpp neigh_request.cpp (+ the other neigh_*.cpp neighbor.cpp points to) comm.cpp
//   procmap.cpp domain.cpp atom_vec.cpp atom_vec_atomic.cpp group.cpp memory.cpp error.cpp universe.cpp STUBS/mpi.c
//
// What cannot be compiled unmodified (and is therefore NOT part of this binary): atom.cpp, force.cpp, update.cpp,
// modify.cpp, output.cpp, lammps.cpp - each includes a committed style_*.h that names package headers absent from
// the tree (style_atom.h:1 atom_vec_angle.h, style_bond.h:1, style_integrate.h:1 mvv_meso.h, style_compute.h:13,
// style_dump.h:5, style_angle.h:1) - and finish.cpp:572 (pointer compared with an integer).  The driver therefore
// does not construct Atom/Force/Update/Modify/Output: it hands the reference classes zero-filled storage of the
// right size for them and fills in the public data members an input script would have set (the commands of
// example/simple/dp.run in their stock CPU form).  No member function of those five classes is ever
// called: Atom::sort therefore cannot run here (atom->sortfreq = 0, the restatement is compared with its own
// sorting switched off), and the step loop below is this file's transcription of the call order of
// Verlet::setup / Verlet::run (src/verlet.cpp:52-110, 227-314) with the Modify dispatch replaced by direct calls
// of FixNVE.  Every arithmetic statement that runs - RanMars, RanPark, PairDPD::compute, Pair::ev_tally,
// virial_fdotr_compute, Neighbor::setup_bins/bin_atoms/half_bin_newton, the 3d newton stencil, Comm::setup/
// exchange/borders/forward_comm/reverse_comm, AtomVecAtomic pack/unpack, Domain::pbc, FixNVE - is the reference's.
//
// Usage:  ref_lmp rng  mars|park SEED N OUT      N uniform() then N gaussian() draws as raw doubles
//         ref_lmp run  IN OUT                    IN/OUT: raw little-endian records, layout in oracle/ref.py
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "mpi.h"
#include "lammps.h"
#include "atom.h"
#include "atom_vec_atomic.h"
#include "comm.h"
#include "domain.h"
#include "error.h"
#include "fix_nve.h"
#include "force.h"
#include "group.h"
#include "memory.h"
#include "modify.h"
#include "neigh_list.h"
#include "neigh_request.h"
#include "neighbor.h"
#include "output.h"
#include "pair_dpd.h"
#include "random_mars.h"
#include "random_park.h"
#include "universe.h"
#include "update.h"

using namespace LAMMPS_NS;

template <class T> static T *blank()
{
    void *p = calloc(1, sizeof(T));
    return static_cast<T *>(p);
}

// access to what `pair_style dpd T cut seed` / `pair_coeff i j a0 gamma [cut]` store (PairDPD::settings/coeff parse
// their arguments through Force::numeric/bounds, which live in the uncompilable force.cpp)
class PairDPDOpen : public PairDPD {
  public:
    PairDPDOpen(LAMMPS *l) : PairDPD(l) {}
    void style(double T, double cutg, int sd)
    {   // PairDPD::settings src/pair_dpd.cpp:186-208
        temperature = T;
        cut_global = cutg;
        seed = sd;
        random = new RanMars(lmp, seed + comm->me);
    }
    void coeff1(int i, int j, double a, double g, double c)
    {   // PairDPD::coeff src/pair_dpd.cpp:214-241
        if (!allocated) allocate();
        a0[i][j] = a;
        gamma[i][j] = g;
        cut[i][j] = c;
        setflag[i][j] = 1;
    }
    double energy() const { return eng_vdwl; }
    const double *vir() const { return virial; }
};

static void wr(FILE *f, const void *p, size_t n)
{
    if (fwrite(p, 1, n, f) != n) { perror("write"); exit(3); }
}
static void rd(FILE *f, void *p, size_t n)
{
    if (fread(p, 1, n, f) != n) { fprintf(stderr, "short read\n"); exit(3); }
}

static int run_rng(int argc, char **argv)
{
    if (argc < 6) return 2;
    const int seed = atoi(argv[3]), n = atoi(argv[4]);
    LAMMPS *lmp = blank<LAMMPS>();
    lmp->world = MPI_COMM_WORLD;
    std::vector<double> out(2 * (size_t)n);
    if (!strcmp(argv[2], "mars")) {
        RanMars r(lmp, seed);
        for (int i = 0; i < n; i++) out[i] = r.uniform();
        for (int i = 0; i < n; i++) out[n + i] = r.gaussian();
    } else {
        RanPark r(lmp, seed);
        for (int i = 0; i < n; i++) out[i] = r.uniform();
        for (int i = 0; i < n; i++) out[n + i] = r.gaussian();
    }
    FILE *f = fopen(argv[5], "wb");
    if (!f) return 3;
    wr(f, out.data(), out.size() * sizeof(double));
    fclose(f);
    return 0;
}

struct Header {
    int n, ntypes, nsteps, every, seed, ncoeff, nsample, reserved;
    double lo[3], hi[3];
    double temperature, cut_global, skin, dt;
};

static int run_sim(int argc, char **argv)
{
    if (argc < 4) return 2;
    FILE *in = fopen(argv[2], "rb");
    if (!in) { perror(argv[2]); return 3; }
    Header h;
    rd(in, &h, sizeof h);
    std::vector<double> x(3 * (size_t)h.n), v(3 * (size_t)h.n), mass(h.ntypes), coeff(5 * (size_t)h.ncoeff);
    std::vector<int> type(h.n), sample(h.nsample);
    rd(in, x.data(), x.size() * 8);
    rd(in, v.data(), v.size() * 8);
    rd(in, type.data(), type.size() * 4);
    rd(in, mass.data(), mass.size() * 8);
    rd(in, coeff.data(), coeff.size() * 8);      // (i, j, a0, gamma, cut) per pair_coeff line
    rd(in, sample.data(), sample.size() * 4);    // steps after which the state is written (0 = after setup)
    fclose(in);

    int margc = 0;
    char **margv = NULL;
    MPI_Init(&margc, &margv);

    LAMMPS *lmp = blank<LAMMPS>();
    lmp->world = MPI_COMM_WORLD;
    lmp->screen = NULL;
    lmp->logfile = NULL;
    lmp->infile = NULL;
    lmp->memory = new Memory(lmp);
    lmp->error = new Error(lmp);
    lmp->universe = new Universe(lmp, MPI_COMM_WORLD);
    Atom *atom = lmp->atom = blank<Atom>();
    Force *force = lmp->force = blank<Force>();
    Update *update = lmp->update = blank<Update>();
    Modify *modify = lmp->modify = blank<Modify>();
    Output *output = lmp->output = blank<Output>();
    (void)output;
    lmp->group = new Group(lmp);
    Domain *domain = lmp->domain = new Domain(lmp);
    Comm *comm = lmp->comm = new Comm(lmp);
    Neighbor *neighbor = lmp->neighbor = new Neighbor(lmp);

    // units lj (Update::set_units src/update.cpp "lj" branch): every conversion factor is 1
    force->boltz = 1.0; force->hplanck = 0.18292026; force->mvv2e = 1.0; force->ftm2v = 1.0; force->mv2d = 1.0;
    force->nktv2p = 1.0; force->qqr2e = 1.0; force->qe2f = 1.0; force->vxmu2f = 1.0; force->xxt2kmu = 1.0;
    force->e_mass = 0.0; force->hhmrr2e = 0.0; force->mvh2r = 0.0; force->angstrom = 1.0; force->femtosecond = 1.0;
    force->qelectron = 1.0; force->dielectric = 1.0; force->qqrd2e = 1.0;
    // Force::Force defaults (src/force.cpp:50-60): newton on, special_lj = {1, 0, 0, 0}
    force->newton = force->newton_pair = force->newton_bond = 1;
    force->special_lj[0] = force->special_coul[0] = 1.0;
    update->dt = h.dt;
    update->ntimestep = 0;
    update->whichflag = 1;
    static char integrate_style[] = "verlet";
    update->integrate_style = integrate_style;
    modify->nfix = 0;
    modify->ncompute = 0;

    // atom_style atomic; read_data: box, atoms, velocities, masses (Atom::Atom defaults src/atom.cpp:60-170)
    atom->ntypes = h.ntypes;
    atom->natoms = h.n;
    atom->nlocal = atom->nghost = atom->nmax = 0;
    atom->tag_enable = 1;
    atom->map_style = 0;
    atom->sortfreq = 0;                 // Atom::sort is in atom.cpp: not available (see header)
    atom->molecular = 0;
    AtomVecAtomic *avec = new AtomVecAtomic(lmp);
    atom->avec = avec;
    lmp->memory->create(atom->mass, h.ntypes + 1, "atom:mass");
    lmp->memory->create(atom->mass_setflag, h.ntypes + 1, "atom:mass_setflag");
    for (int t = 1; t <= h.ntypes; t++) { atom->mass[t] = mass[t - 1]; atom->mass_setflag[t] = 1; }

    domain->dimension = 3;
    for (int d = 0; d < 3; d++) { domain->boxlo[d] = h.lo[d]; domain->boxhi[d] = h.hi[d]; }
    domain->set_initial_box();          // as ReadData::command src/read_data.cpp:186-190
    domain->set_global_box();
    comm->set_proc_grid();
    domain->set_local_box();

    avec->grow(h.n);
    for (int i = 0; i < h.n; i++) {
        // AtomVecAtomic::data_atom src/atom_vec_atomic.cpp (tag, type, x, default image, mask 1, v = 0) + Velocities
        atom->tag[i] = i + 1;
        atom->type[i] = type[i];
        for (int d = 0; d < 3; d++) { atom->x[i][d] = x[3 * (size_t)i + d]; atom->v[i][d] = v[3 * (size_t)i + d]; }
        atom->image[i] = ((tagint)IMGMAX << IMG2BITS) | ((tagint)IMGMAX << IMGBITS) | IMGMAX;
        atom->mask[i] = 1;
    }
    atom->nlocal = h.n;

    comm->ghost_velocity = 1;           // communicate single vel yes
    neighbor->skin = h.skin;            // neighbor SKIN bin
    neighbor->style = 1;
    neighbor->every = h.every;          // neigh_modify delay 0 every E check no
    neighbor->delay = 0;
    neighbor->dist_check = 0;

    PairDPDOpen *pair = new PairDPDOpen(lmp);
    force->pair = pair;
    pair->style(h.temperature, h.cut_global, h.seed);
    for (int c = 0; c < h.ncoeff; c++) {
        const double *q = &coeff[5 * (size_t)c];
        pair->coeff1((int)q[0], (int)q[1], q[2], q[3], q[4] > 0.0 ? q[4] : h.cut_global);
    }

    static char a0[] = "1", a1[] = "all", a2[] = "nve";
    char *fixarg[3] = {a0, a1, a2};
    FixNVE *nve = new FixNVE(lmp, 3, fixarg);

    // LAMMPS::init order src/lammps.cpp (force, domain, neighbor, comm); fix init as Modify::init does
    pair->init();
    domain->init();
    atom->firstgroup = -1;              // Atom::init src/atom.cpp:384-390 (no atom_modify first), then avec->init()
    avec->init();
    nve->init();
    neighbor->init();
    comm->init();

    FILE *out = fopen(argv[3], "wb");
    if (!out) { perror(argv[3]); return 3; }
    std::vector<double> rec;
    auto dump = [&](int step) {
        const int n = atom->nlocal;
        rec.assign(9 * (size_t)h.n, 0.0);
        for (int i = 0; i < n; i++) {
            const size_t t = (size_t)atom->tag[i] - 1;
            for (int d = 0; d < 3; d++) {
                rec[3 * t + d] = atom->x[i][d];
                rec[3 * (size_t)h.n + 3 * t + d] = atom->v[i][d];
                rec[6 * (size_t)h.n + 3 * t + d] = atom->f[i][d];
            }
        }
        long long nneigh = 0;
        NeighList *l = neighbor->lists[0];
        for (int ii = 0; ii < l->inum; ii++) nneigh += l->numneigh[l->ilist[ii]];
        double scal[12] = {(double)step, (double)n, (double)atom->nghost, (double)nneigh, pair->energy(), 0, 0, 0, 0, 0, 0,
                           (double)neighbor->ncalls};
        for (int k = 0; k < 6; k++) scal[5 + k] = pair->vir()[k];
        wr(out, scal, sizeof scal);
        wr(out, rec.data(), rec.size() * 8);
    };
    auto force_clear = [&]() {   // Verlet::force_clear src/verlet.cpp:345-365 (newton on: ghosts too)
        const int nall = atom->nlocal + atom->nghost;
        for (int i = 0; i < nall; i++) atom->f[i][0] = atom->f[i][1] = atom->f[i][2] = 0.0;
    };

    // Verlet::setup src/verlet.cpp:52-110
    domain->pbc();
    domain->reset_box();
    comm->setup();
    neighbor->setup_bins();
    comm->exchange();
    comm->borders();
    neighbor->build();
    neighbor->ncalls = 0;
    force_clear();
    pair->compute(1, 2);                // eflag global, vflag global via F dot r (Integrate::ev_set on a thermo step)
    comm->reverse_comm();
    int si = 0;
    if (si < h.nsample && sample[si] == 0) { dump(0); si++; }

    // Verlet::run src/verlet.cpp:227-314
    for (int s = 1; s <= h.nsteps; s++) {
        update->ntimestep = s;
        const bool out_now = si < h.nsample && sample[si] == s;
        nve->initial_integrate(0);
        if (neighbor->decide() == 0) comm->forward_comm();
        else {
            domain->pbc();
            comm->exchange();
            comm->borders();
            neighbor->build();
        }
        force_clear();
        if (out_now) pair->compute(1, 2);
        else pair->compute(0, 0);
        comm->reverse_comm();
        nve->final_integrate();
        if (out_now) { dump(s); si++; }
    }
    fclose(out);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc >= 2 && !strcmp(argv[1], "rng")) return run_rng(argc, argv);
    if (argc >= 2 && !strcmp(argv[1], "run")) return run_sim(argc, argv);
    fprintf(stderr, "usage: ref_lmp rng mars|park SEED N OUT | ref_lmp run IN OUT\n");
    return 2;
}
