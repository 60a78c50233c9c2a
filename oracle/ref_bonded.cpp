// oracle/_ref driver for the reference's own BONDED styles: src/MOLECULE/bond_harmonic.cpp, bond_fene.cpp and
// angle_harmonic.cpp (with src/bond.cpp, src/angle.cpp) compiled UNMODIFIED where they lie (oracle/build_ref.sh) and run on a
// hand-filled neighbor->bondlist / anglelist.  TEST INFRASTRUCTURE ONLY: pins the bonded terms of oracle/meso_sim.py, which are
// what the HIP kernels (bond.hip) are compared with.
//
// As in ref_harness.cpp the classes whose .cpp cannot be compiled unmodified (Atom, Force, Neighbor is fine but not needed,
// Update, Comm) are zero-filled storage with the public members set that `atom_style bond|angle`, `bond_style`, `bond_coeff`,
// `angle_style`, `angle_coeff` would have set; coefficients are written into the styles' own arrays through a derived class
// (Bond*::coeff parses text through Force::numeric/bounds, which live in the uncompilable force.cpp).  No ghost atoms: the
// test decks keep every bond and angle away from the periodic boundary, so x[i] - x[j] needs no image.
//
// Usage: ref_bonded IN OUT      IN: int n, nbond, nangle, nbt, nat, style (0 harmonic, 1 fene); double x[n][3]; int bond[nbond][3]
//                                   (atom, atom, type; 0-based atoms, 1-based type); double bcoef[nbt][4]; int angle[nangle][4];
//                                   double acoef[nat][2] (K, theta0 in degrees)
//                               OUT: double f_bond[n][3], e_bond, f_angle[n][3], e_angle
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "mpi.h"
#include "lammps.h"
#include "angle_harmonic.h"
#include "atom.h"
#include "bond_fene.h"
#include "bond_harmonic.h"
#include "comm.h"
#include "error.h"
#include "force.h"
#include "math_const.h"
#include "memory.h"
#include "neighbor.h"
#include "universe.h"
#include "update.h"

using namespace LAMMPS_NS;

template <class T> static T *blank() { return static_cast<T *>(calloc(1, sizeof(T))); }

class BondHarmonicOpen : public BondHarmonic {
  public:
    BondHarmonicOpen(LAMMPS *l) : BondHarmonic(l) {}
    void set(int t, double kk, double rr)      // BondHarmonic::coeff src/MOLECULE/bond_harmonic.cpp:113-133
    {
        if (!allocated) allocate();
        k[t] = kk; r0[t] = rr; setflag[t] = 1;
    }
};
class BondFENEOpen : public BondFENE {
  public:
    BondFENEOpen(LAMMPS *l) : BondFENE(l) {}
    void set(int t, double kk, double rr, double ee, double ss)      // BondFENE::coeff src/MOLECULE/bond_fene.cpp:152-176
    {
        if (!allocated) allocate();
        k[t] = kk; r0[t] = rr; epsilon[t] = ee; sigma[t] = ss; setflag[t] = 1;
    }
};
class AngleHarmonicOpen : public AngleHarmonic {
  public:
    AngleHarmonicOpen(LAMMPS *l) : AngleHarmonic(l) {}
    void set(int t, double kk, double th_deg)      // AngleHarmonic::coeff src/MOLECULE/angle_harmonic.cpp:161-183: degrees -> radians
    {
        if (!allocated) allocate();
        k[t] = kk; theta0[t] = th_deg / 180.0 * MathConst::MY_PI; setflag[t] = 1;
    }
};

static void rd(FILE *f, void *p, size_t n)
{
    if (n && fread(p, 1, n, f) != n) { fprintf(stderr, "short read\n"); exit(3); }
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: ref_bonded IN OUT\n"); return 2; }
    FILE *in = fopen(argv[1], "rb");
    if (!in) { perror(argv[1]); return 3; }
    int h[6];
    rd(in, h, sizeof h);
    const int n = h[0], nb = h[1], na = h[2], nbt = h[3], nat = h[4], style = h[5];
    std::vector<double> x(3 * (size_t)n), bc(4 * (size_t)nbt), ac(2 * (size_t)nat);
    std::vector<int> bl(3 * (size_t)nb), al(4 * (size_t)na);
    rd(in, x.data(), x.size() * 8);
    rd(in, bl.data(), bl.size() * 4);
    rd(in, bc.data(), bc.size() * 8);
    rd(in, al.data(), al.size() * 4);
    rd(in, ac.data(), ac.size() * 8);
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
    Comm *comm = lmp->comm = blank<Comm>();
    Neighbor *neighbor = lmp->neighbor = blank<Neighbor>();
    (void)update;
    comm->nthreads = 1;
    force->newton = force->newton_pair = force->newton_bond = 1;      // Force::Force defaults, src/force.cpp:50-60
    atom->nlocal = n; atom->nghost = 0; atom->nmax = n; atom->natoms = n;
    atom->nbondtypes = nbt; atom->nangletypes = nat;
    lmp->memory->create(atom->x, n, 3, "atom:x");
    lmp->memory->create(atom->f, n, 3, "atom:f");
    for (int i = 0; i < n; i++)
        for (int d = 0; d < 3; d++) atom->x[i][d] = x[3 * (size_t)i + d];
    // Neighbor::bond_all / angle_all (src/neigh_bond.cpp) would fill these from the per-atom topology: here by hand
    lmp->memory->create(neighbor->bondlist, nb > 0 ? nb : 1, 3, "neigh:bondlist");
    lmp->memory->create(neighbor->anglelist, na > 0 ? na : 1, 4, "neigh:anglelist");
    neighbor->nbondlist = nb; neighbor->nanglelist = na;
    for (int k = 0; k < nb; k++)
        for (int d = 0; d < 3; d++) neighbor->bondlist[k][d] = bl[3 * (size_t)k + d];
    for (int k = 0; k < na; k++)
        for (int d = 0; d < 4; d++) neighbor->anglelist[k][d] = al[4 * (size_t)k + d];

    FILE *out = fopen(argv[2], "wb");
    if (!out) { perror(argv[2]); return 3; }
    auto clear = [&]() { for (int i = 0; i < n; i++) atom->f[i][0] = atom->f[i][1] = atom->f[i][2] = 0.0; };
    auto dump = [&](double e) {
        for (int i = 0; i < n; i++) fwrite(atom->f[i], 8, 3, out);
        fwrite(&e, 8, 1, out);
    };
    clear();
    double eb = 0.0;
    if (nb > 0) {
        if (style == 0) {
            BondHarmonicOpen b(lmp);
            for (int t = 1; t <= nbt; t++) b.set(t, bc[4 * (t - 1)], bc[4 * (t - 1) + 1]);
            b.compute(1, 0);
            eb = b.energy;
        } else {
            BondFENEOpen b(lmp);
            for (int t = 1; t <= nbt; t++) b.set(t, bc[4 * (t - 1)], bc[4 * (t - 1) + 1], bc[4 * (t - 1) + 2], bc[4 * (t - 1) + 3]);
            b.compute(1, 0);
            eb = b.energy;
        }
    }
    dump(eb);
    clear();
    double ea = 0.0;
    if (na > 0) {
        AngleHarmonicOpen a(lmp);
        for (int t = 1; t <= nat; t++) a.set(t, ac[2 * (t - 1)], ac[2 * (t - 1) + 1]);
        a.compute(1, 0);
        ea = a.energy;
    }
    dump(ea);
    fclose(out);
    return 0;
}
