// Test harness (CPU only, compiled by tests/test_glue_restart.py where the reference tree is mounted): the glue's
// MesoHipPairDPD::write_restart / read_restart (lammps_glue/meso_hip_glue.cpp) executed against LAMMPS' own Pair base class
// (src/pair.cpp, compiled unmodified) so that the bytes they emit can be compared with the record MesoPairDPD::write_restart
// emits for the same coefficients (src/USER-MESO/pair_dpd_meso.cu:363-447: settings = cut_global f64, seed i32, mix_flag i32;
// then per i <= j: setflag i32 and, if set, a0 gamma sigma expw cut as f64).  The C ABI is a recording stub
// (tests/c/meso_stub.c): no GPU, no product code path involved.
//
//   glue_restart write OUT        pair_style dpd/meso 1.0 419084618; coefficients 1-1 and 1-2 set, 2-2 not set
//   glue_restart read  IN  LOG    read_restart(IN); every meso_* call the glue makes is appended to LOG by the stub
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "mpi.h"
#include "lammps.h"
#include "atom.h"
#include "comm.h"
#include "error.h"
#include "force.h"
#include "memory.h"
#include "universe.h"
#include "update.h"
#include "meso_hip_glue.h"

using namespace LAMMPS_NS;

template <class T> static T *blank() { return static_cast<T *>(calloc(1, sizeof(T))); }

class PairOpen : public MesoHipPairDPD {
  public:
    PairOpen(LAMMPS *l) : MesoHipPairDPD(l) {}
    void fill()
    {
        cut_global = 1.0;
        seed = 419084618;
        allocate();
        const double v[2][5] = {{15.0, 4.5, 3.0, 1.0, 1.0}, {40.0, 6.0, 3.4641016151377544, 0.5, 1.25}};
        int q = 0;
        for (int i = 1; i <= 2; i++)
            for (int j = i; j <= 2; j++) {
                if (i == 2 && j == 2) continue;      // 2-2 left unset: the record then holds the flag alone
                setflag[i][j] = 1;
                a0[i][j] = v[q][0]; gamma[i][j] = v[q][1]; sigma[i][j] = v[q][2]; expw[i][j] = v[q][3]; cut[i][j] = v[q][4];
                q++;
            }
    }
};

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    int margc = 0;
    char **margv = NULL;
    MPI_Init(&margc, &margv);
    LAMMPS *lmp = blank<LAMMPS>();
    lmp->world = MPI_COMM_WORLD;
    lmp->memory = new Memory(lmp);
    lmp->error = new Error(lmp);
    lmp->universe = new Universe(lmp, MPI_COMM_WORLD);
    Atom *atom = lmp->atom = blank<Atom>();
    lmp->force = blank<Force>();
    lmp->update = blank<Update>();
    Comm *comm = lmp->comm = blank<Comm>();
    comm->me = 0; comm->nprocs = 1; comm->nthreads = 1;
    atom->ntypes = 2;
    lmp->memory->create(atom->mass, 3, "atom:mass");
    atom->mass[1] = 1.0; atom->mass[2] = 2.0;
    PairOpen p(lmp);
    if (!strcmp(argv[1], "write")) {
        p.fill();
        FILE *fp = fopen(argv[2], "wb");
        if (!fp) return 3;
        p.write_restart(fp);
        fclose(fp);
        return 0;
    }
    if (!strcmp(argv[1], "read") && argc >= 4) {
        setenv("MESO_STUB_LOG", argv[3], 1);
        FILE *fp = fopen(argv[2], "rb");
        if (!fp) return 3;
        p.read_restart(fp);
        // what was read must write back to the same bytes
        FILE *fo = fopen((std::string(argv[3]) + ".rewrite").c_str(), "wb");
        p.write_restart(fo);
        fclose(fo);
        fclose(fp);
        return 0;
    }
    return 2;
}
