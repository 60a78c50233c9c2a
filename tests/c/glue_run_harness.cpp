// Test harness: the LAMMPS-side binding (lammps_glue/meso_hip_glue.cpp) executed against the REAL libmeso_hip.so on a GPU.
//
// Built in this container only (tests/c/build_glue_run.sh: the glue and this driver compiled against the reference's own headers,
// linked with the reference's unmodified base classes - src/pair.cpp, fix.cpp, compute.cpp, integrate.cpp, group.cpp, domain.cpp,
// comm.cpp ... as oracle/build_ref.sh compiles them - and with meso_amd/libmeso_hip.so); the binary travels to the GPU box, the
// reference tree does not.  TEST INFRASTRUCTURE: nothing of the product links or loads it.
//
// What runs: the style objects a LAMMPS input deck would create - MesoHipPairDPD(Fast)::settings / coeff / init_one / compute,
// MesoHipFixNVE::initial_integrate / final_integrate, MesoHipComputeTemp::compute_scalar, MesoHipIntegrate::upload / download
// (the counterparts of pair_dpd_meso.h:30-41, fix_nve_meso.cu:97-199, compute_temp_meso.cu:77-101, mvv_meso.cu:139-219) - called
// through the virtuals of the reference's base classes (Pair *, Fix *, Compute *), in the order of a host-driven timestep
// (mvv_meso.cu:243-425: initial_integrate, decide / reneighbor or forward_comm, force_clear, compute, final_integrate).  Atom,
// Force, Update, Modify and Output are zero-filled storage with the data members a deck would have set (their .cpp files include
// style headers of packages that are not in the tree, see oracle/ref_harness.cpp); MesoHipIntegrate::setup()/run() themselves
// call Output::setup / Output::write and are therefore not reached: the driver makes the same library calls around upload() and
// download() that setup() makes.
//
//   glue_run IN OUT      IN: header + x, v (see Header); OUT: n, then tag[n], x[3n], v[3n], f[3n], temperature after every step
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
#include "force.h"
#include "group.h"
#include "memory.h"
#include "modify.h"
#include "neighbor.h"
#include "output.h"
#include "universe.h"
#include "update.h"
#include "meso_hip_glue.h"

using namespace LAMMPS_NS;

template <class T> static T *blank() { return static_cast<T *>(calloc(1, sizeof(T))); }

struct Header {
    int n, nsteps, every, seed, fast, reserved[3];
    double lo[3], hi[3];
    double cut_global, skin, dt, a0, gamma, sigma;
};

class IntegrateOpen : public MesoHipIntegrate {
  public:
    IntegrateOpen(LAMMPS *l) : MesoHipIntegrate(l, 0, NULL) {}
    void up() { upload(); }
    void down() { download(); }
};

static void die(const char *what) { fprintf(stderr, "glue_run: %s: %s\n", what, meso_last_error()); exit(4); }

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    FILE *in = fopen(argv[1], "rb");
    if (!in) { perror(argv[1]); return 3; }
    Header h;
    if (fread(&h, sizeof h, 1, in) != 1) return 3;
    std::vector<double> x(3 * (size_t)h.n), v(3 * (size_t)h.n);
    if (fread(x.data(), 8, x.size(), in) != x.size() || fread(v.data(), 8, v.size(), in) != v.size()) return 3;
    fclose(in);

    int margc = 0;
    char **margv = NULL;
    MPI_Init(&margc, &margv);
    LAMMPS *lmp = blank<LAMMPS>();
    lmp->world = MPI_COMM_WORLD;
    lmp->memory = new Memory(lmp);
    lmp->error = new Error(lmp);
    lmp->universe = new Universe(lmp, MPI_COMM_WORLD);
    Atom *atom = lmp->atom = blank<Atom>();
    Force *force = lmp->force = blank<Force>();
    Update *update = lmp->update = blank<Update>();
    lmp->modify = blank<Modify>();
    lmp->output = blank<Output>();
    lmp->group = new Group(lmp);
    Domain *domain = lmp->domain = new Domain(lmp);
    Comm *comm = lmp->comm = new Comm(lmp);
    Neighbor *neighbor = lmp->neighbor = new Neighbor(lmp);

    // what the commands of example/simple/dp.run leave in the data members: units lj, atom_style dpd/atomic/meso, read_data,
    // neighbor 0.3 bin, neigh_modify delay 0 every 5 check no, timestep 0.005
    force->boltz = 1.0; force->mvv2e = 1.0; force->ftm2v = 1.0; force->nktv2p = 1.0;
    force->newton = force->newton_pair = force->newton_bond = 0;        // MesoHipIntegrate::init (mvv_meso.cu:101-110)
    force->special_lj[0] = force->special_coul[0] = 1.0;
    update->dt = h.dt;
    update->ntimestep = 0;
    atom->ntypes = 1;
    atom->natoms = h.n;
    atom->tag_enable = 1;
    atom->molecular = 0;
    AtomVecAtomic *avec = new MesoHipAtomVecDPDAtomic(lmp);
    atom->avec = avec;
    lmp->memory->create(atom->mass, 2, "atom:mass");
    lmp->memory->create(atom->mass_setflag, 2, "atom:mass_setflag");
    atom->mass[1] = 1.0; atom->mass_setflag[1] = 1;
    domain->dimension = 3;
    for (int d = 0; d < 3; d++) { domain->boxlo[d] = h.lo[d]; domain->boxhi[d] = h.hi[d]; }
    domain->set_initial_box();
    domain->set_global_box();
    comm->set_proc_grid();
    domain->set_local_box();
    avec->grow(h.n);
    for (int i = 0; i < h.n; i++) {
        for (int d = 0; d < 3; d++) { atom->x[i][d] = x[3 * (size_t)i + d]; atom->v[i][d] = v[3 * (size_t)i + d]; atom->f[i][d] = 0.0; }
        atom->tag[i] = i + 1; atom->type[i] = 1; atom->mask[i] = 1;
        atom->image[i] = (512 << 20) | (512 << 10) | 512;
    }
    atom->nlocal = h.n;
    neighbor->skin = h.skin; neighbor->every = h.every; neighbor->delay = 0; neighbor->dist_check = 0;

    // the style objects of the deck, held by pointers to the reference's base classes
    char s_rc[32], s_seed[32], s_a0[32], s_g[32], s_s[32];
    snprintf(s_rc, sizeof s_rc, "%.17g", h.cut_global); snprintf(s_seed, sizeof s_seed, "%d", h.seed);
    snprintf(s_a0, sizeof s_a0, "%.17g", h.a0); snprintf(s_g, sizeof s_g, "%.17g", h.gamma); snprintf(s_s, sizeof s_s, "%.17g", h.sigma);
    Pair *pair = h.fast ? (Pair *)new MesoHipPairDPDFast(lmp) : (Pair *)new MesoHipPairDPD(lmp);
    force->pair = pair;
    char *sarg[] = {s_rc, s_seed};
    pair->settings(2, sarg);                                   // pair_style dpd/meso 1.0 419084618
    char one[] = "1", star[] = "*", expw[] = "1.0";
    char *carg[] = {one, star, s_a0, s_g, s_s, expw};
    pair->coeff(6, carg);                                      // pair_coeff 1 * 15 4.5 3.0 1.0
    if (pair->init_one(1, 1) != h.cut_global) { fprintf(stderr, "glue_run: init_one\n"); return 4; }
    char f0[] = "3", all[] = "all", f2[] = "nve/meso";
    char *farg[] = {f0, all, f2};
    Fix *nve = new MesoHipFixNVE(lmp, 3, farg);                // fix 3 all nve/meso
    if (nve->setmask() != (FixConst::INITIAL_INTEGRATE | FixConst::FINAL_INTEGRATE)) return 4;
    char c0[] = "mythermo", c2[] = "temp/meso";
    char *targ[] = {c0, all, c2};
    Compute *temp = new MesoHipComputeTemp(lmp, 3, targ);      // compute mythermo all temp/meso
    IntegrateOpen integ(lmp);

    meso_ctx *ctx = MesoHipContext::get(lmp);
    // MesoHipIntegrate::setup without Output::setup: upload, step number, setup forces, download
    integ.up();
    if (meso_step_advance(ctx, update->ntimestep)) die("step_advance");
    if (meso_setup(ctx)) die("setup");
    integ.down();

    FILE *out = fopen(argv[2], "wb");
    if (!out) return 3;
    fwrite(&h.n, 4, 1, out);
    std::vector<double> temps;
    for (int it = 0; it < h.nsteps; it++) {
        update->ntimestep++;
        if (meso_step_advance(ctx, update->ntimestep)) die("step_advance");
        nve->initial_integrate(0);
        int rebuild = 0;
        if (meso_neighbor_decide(ctx, &rebuild)) die("decide");
        if (rebuild) { if (meso_reneighbor(ctx)) die("reneighbor"); }
        else if (meso_halo_forward(ctx)) die("halo_forward");
        if (meso_force_clear(ctx, MESO_RANGE_LOCAL)) die("force_clear");
        pair->compute(0, 0);
        nve->final_integrate();
        temps.push_back(temp->compute_scalar());
    }
    integ.down();
    const int n = atom->nlocal;
    fwrite(atom->tag, 4, n, out);
    fwrite(atom->x[0], 8, 3 * (size_t)n, out);
    fwrite(atom->v[0], 8, 3 * (size_t)n, out);
    fwrite(atom->f[0], 8, 3 * (size_t)n, out);
    fwrite(temps.data(), 8, temps.size(), out);
    fclose(out);
    meso_finalize(ctx);
    return n == h.n ? 0 : 5;
}
