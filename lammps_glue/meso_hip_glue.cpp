/* Implementation of the LAMMPS-side binding declared in meso_hip_glue.h.  Pure host C++ against LAMMPS'
   own headers; links with -lmeso_hip.  See INTEGRATION.md. */

#include "string.h"
#include "stdlib.h"
#include "meso_hip_glue.h"
#include "atom.h"
#include "atom_vec.h"
#include "comm.h"
#include "domain.h"
#include "error.h"
#include "force.h"
#include "memory.h"
#include "modify.h"
#include "neighbor.h"
#include "output.h"
#include "universe.h"
#include "update.h"

using namespace LAMMPS_NS;

#define MESO(call) MesoHipContext::check(lmp, (call), FLERR)

/* type ranges of the *_coeff commands: "n", "*", "n*", "*m", "n*m" over 1..nmax - the grammar of Force::bounds (src/force.cpp), parsed
   here so that the glue needs nothing of class Force beyond its data members */
static void type_bounds(LAMMPS *lmp, const char *str, int nmax, int &nlo, int &nhi)
{
  const char *star = strchr(str, '*');
  const int n = (int) strlen(str);
  if (!star) nlo = nhi = atoi(str);
  else if (n == 1) { nlo = 1; nhi = nmax; }
  else if (star == str) { nlo = 1; nhi = atoi(str + 1); }
  else if (star == str + n - 1) { nlo = atoi(str); nhi = nmax; }
  else { nlo = atoi(str); nhi = atoi(star + 1); }
  if (nlo < 1 || nhi > nmax || nlo > nhi) lmp->error->all(FLERR, "Numeric index is out of bounds");
}

static meso_ctx *g_ctx = NULL;

meso_ctx *MesoHipContext::get(LAMMPS *lmp)
{
  if (!g_ctx) {
    /* one GPU per rank on the node, like the -device flag of the reference */
    int rc = meso_init(-(lmp->universe->me), &g_ctx);
    check(lmp, rc, FLERR);
    Comm *comm = lmp->comm;
    if (comm->nprocs > 1) {
      /* Several MPI ranks: the library's device-side ghost exchange takes the place of MesoComm::borders/exchange and
         Comm::forward_comm (comm_meso.cu:41-186,256-550).  It must see LAMMPS' own decomposition: the brick grid
         comm->procgrid and this rank's cell comm->myloc (Comm::set_proc_grid, src/comm.cpp:173-300; the library numbers
         the cells x fastest).  Styles are created after the box exists (read_data / create_box call set_proc_grid),
         so procgrid is final here.  The ncclUniqueId is created by MPI rank 0 and broadcast over `world`. */
      if (comm->procgrid[0] * comm->procgrid[1] * comm->procgrid[2] != comm->nprocs)
        lmp->error->all(FLERR, "<MESO> processor grid is not set: define the simulation box before any */meso style");
      if (!comm->uniform)
        lmp->error->all(FLERR, "<MESO> non-uniform processor sub-domains (balance) are not supported");
      unsigned char uid[128];
      memset(uid, 0, sizeof uid);
      if (comm->me == 0) check(lmp, meso_comm_get_unique_id(uid, sizeof uid), FLERR);
      MPI_Bcast(uid, (int) sizeof uid, MPI_BYTE, 0, lmp->world);
      int grid_rank = comm->myloc[0] + comm->procgrid[0] * (comm->myloc[1] + comm->procgrid[1] * comm->myloc[2]);
      rc = meso_comm_init(g_ctx, comm->nprocs, grid_rank, comm->procgrid, MESO_TRANSPORT_RCCL, uid, sizeof uid);
      check(lmp, rc, FLERR);
    }
  }
  return g_ctx;
}

void MesoHipContext::check(LAMMPS *lmp, int rc, const char *file, int line)
{
  if (rc) lmp->error->one(file, line, meso_last_error());
}

/* ---------------------------------------------------------------------- pair */

MesoHipPairDPD::MesoHipPairDPD(LAMMPS *lmp) : Pair(lmp)
{
  split_flag = 1;
  style_id = MESO_PAIR_DPD;
  cut = a0 = gamma = sigma = expw = NULL;
}

MesoHipPairDPDFast::MesoHipPairDPDFast(LAMMPS *lmp) : MesoHipPairDPD(lmp) { style_id = MESO_PAIR_DPD_FAST; }

MesoHipPairDPDMini::MesoHipPairDPDMini(LAMMPS *lmp) : MesoHipPairDPD(lmp) { style_id = MESO_PAIR_DPD_MINI; }

void MesoHipPairDPDMini::settings(int narg, char **arg)
{
  if (narg != 2) error->all(FLERR, "Illegal pair_style command");
  cut_global = 1.0;                    /* MesoPairDPDMini::coeff: cutsq = 1 for every pair */
  seed = atoi(arg[1]);
  MESO(meso_pair_dpd_settings(MesoHipContext::get(lmp), style_id, cut_global, seed));
}

void MesoHipPairDPDMini::coeff(int narg, char **arg)
{
  if (narg < 5) error->all(FLERR, "Incorrect args for pair coefficients");
  char one[] = "1.0";
  char *full[7] = {arg[0], arg[1], arg[2], arg[3], arg[4], one, one};
  MesoHipPairDPD::coeff(7, full);      /* a0, gamma, sigma are scalars of the style: the library applies them to all pairs */
}

MesoHipPairDPDPolyForce::MesoHipPairDPDPolyForce(LAMMPS *lmp) : MesoHipPairDPD(lmp), gamma_f(NULL), sigma_f(NULL), cut_f(NULL)
{
  style_id = MESO_PAIR_DPD_POLYFORCE;
}

/* the per-pair arrays the polyforce / tableforce styles keep on the host (what their restart records hold) */
static void alloc_gs(LAMMPS *lmp, Memory *memory, int n, int **&setflag, double **&cutsq, double **&cut, float **&g, float **&s,
                     float **&c)
{
  memory->create(setflag, n + 1, n + 1, "pair:setflag");
  memory->create(cutsq, n + 1, n + 1, "pair:cutsq");
  memory->create(cut, n + 1, n + 1, "pair:cut");
  memory->create(g, n + 1, n + 1, "pair:gamma");
  memory->create(s, n + 1, n + 1, "pair:sigma");
  memory->create(c, n + 1, n + 1, "pair:cut_f");
  for (int i = 1; i <= n; i++)
    for (int j = i; j <= n; j++) setflag[i][j] = 0;
  MesoHipContext::check(lmp, meso_set_mass(MesoHipContext::get(lmp), n, lmp->atom->mass), FLERR);
}

/* MesoPairDPDPolyForce::write_restart / MesoPairDPDTableForce::write_restart (identical record layout) */
static void write_gs(FILE *fp, int n, int **setflag, float **g, float **s, float **c)
{
  for (int i = 1; i <= n; i++)
    for (int j = i; j <= n; j++) {
      fwrite(&setflag[i][j], sizeof(int), 1, fp);
      if (setflag[i][j]) {
        fwrite(&g[i][j], sizeof(float), 1, fp);
        fwrite(&s[i][j], sizeof(float), 1, fp);
        fwrite(&c[i][j], sizeof(float), 1, fp);
      }
    }
}

static void read_gs(FILE *fp, int me, MPI_Comm world, int n, int **setflag, double **cut, float **g, float **s, float **c)
{
  for (int i = 1; i <= n; i++)
    for (int j = i; j <= n; j++) {
      if (me == 0) fread(&setflag[i][j], sizeof(int), 1, fp);
      MPI_Bcast(&setflag[i][j], 1, MPI_INT, 0, world);
      if (setflag[i][j]) {
        float v[3];
        if (me == 0) fread(v, sizeof(float), 3, fp);
        MPI_Bcast(v, 3, MPI_FLOAT, 0, world);
        g[i][j] = v[0]; s[i][j] = v[1]; c[i][j] = v[2];
        cut[i][j] = v[2];
        setflag[i][j] = 0;   /* the polynomial / table is not in the file: pair_coeff must set the pair again */
      }
    }
}

static void write_gs_settings(FILE *fp, double cut_global, int seed, int mix_flag)
{
  float cg = (float) cut_global;
  fwrite(&cg, sizeof(float), 1, fp);
  fwrite(&seed, sizeof(int), 1, fp);
  fwrite(&mix_flag, sizeof(int), 1, fp);
}

static void read_gs_settings(FILE *fp, int me, MPI_Comm world, double &cut_global, int &seed, int &mix_flag)
{
  float cg = 0.0f;
  if (me == 0) {
    fread(&cg, sizeof(float), 1, fp);
    fread(&seed, sizeof(int), 1, fp);
    fread(&mix_flag, sizeof(int), 1, fp);
  }
  MPI_Bcast(&cg, 1, MPI_FLOAT, 0, world);
  MPI_Bcast(&seed, 1, MPI_INT, 0, world);
  MPI_Bcast(&mix_flag, 1, MPI_INT, 0, world);
  cut_global = cg;
}

void MesoHipPairDPDPolyForce::allocate_gs()
{
  allocated = 1;
  alloc_gs(lmp, memory, atom->ntypes, setflag, cutsq, cut, gamma_f, sigma_f, cut_f);
}

void MesoHipPairDPDPolyForce::write_restart(FILE *fp)
{
  write_restart_settings(fp);
  write_gs(fp, atom->ntypes, setflag, gamma_f, sigma_f, cut_f);
}

void MesoHipPairDPDPolyForce::read_restart(FILE *fp)
{
  read_restart_settings(fp);
  if (!allocated) allocate_gs();
  read_gs(fp, comm->me, world, atom->ntypes, setflag, cut, gamma_f, sigma_f, cut_f);
  error->warning(FLERR, "PolyForce polynomial not loaded in read_restart, please set using the pair_coeff command");
}

void MesoHipPairDPDPolyForce::write_restart_settings(FILE *fp) { write_gs_settings(fp, cut_global, seed, mix_flag); }

void MesoHipPairDPDPolyForce::read_restart_settings(FILE *fp)
{
  read_gs_settings(fp, comm->me, world, cut_global, seed, mix_flag);
  MESO(meso_pair_dpd_settings(MesoHipContext::get(lmp), style_id, cut_global, seed));
}

void MesoHipPairDPDPolyForce::coeff(int narg, char **arg)
{
  if (narg < 6 || narg != 6 + atoi(arg[4])) error->all(FLERR, "Incorrect args for pair coefficients");
  int n = atom->ntypes;
  if (!allocated) allocate_gs();
  int ilo, ihi, jlo, jhi;
  type_bounds(lmp, arg[0], n, ilo, ihi);
  type_bounds(lmp, arg[1], n, jlo, jhi);
  int order = atoi(arg[4]);
  double *c = new double[order + 1];
  for (int k = 0; k <= order; k++) c[k] = atof(arg[5 + k]);
  int count = 0;
  for (int i = ilo; i <= ihi; i++)
    for (int j = MAX(jlo, i); j <= jhi; j++) {
      MESO(meso_pair_dpd_polyforce_coeff(MesoHipContext::get(lmp), i, j, atof(arg[2]), atof(arg[3]), order, c));
      gamma_f[i][j] = (float) atof(arg[2]); sigma_f[i][j] = (float) atof(arg[3]); cut_f[i][j] = (float) cut_global;
      cut[i][j] = cut_global;
      setflag[i][j] = 1;
      count++;
    }
  delete [] c;
  if (count == 0) error->all(FLERR, "Incorrect args for pair coefficients");
}

MesoHipPairDPDTableForce::MesoHipPairDPDTableForce(LAMMPS *lmp) : MesoHipPairDPD(lmp), table_length(0), gamma_f(NULL), sigma_f(NULL), cut_f(NULL)
{
  style_id = MESO_PAIR_DPD_TABLEFORCE;
}

void MesoHipPairDPDTableForce::allocate_gs()
{
  allocated = 1;
  alloc_gs(lmp, memory, atom->ntypes, setflag, cutsq, cut, gamma_f, sigma_f, cut_f);
}

void MesoHipPairDPDTableForce::write_restart(FILE *fp)
{
  write_restart_settings(fp);
  write_gs(fp, atom->ntypes, setflag, gamma_f, sigma_f, cut_f);
}

void MesoHipPairDPDTableForce::read_restart(FILE *fp)
{
  read_restart_settings(fp);
  if (!allocated) allocate_gs();
  read_gs(fp, comm->me, world, atom->ntypes, setflag, cut, gamma_f, sigma_f, cut_f);
  error->warning(FLERR, "Force table not loaded in read_restart, please set using the pair_style and pair_coeff commands");
}

void MesoHipPairDPDTableForce::write_restart_settings(FILE *fp) { write_gs_settings(fp, cut_global, seed, mix_flag); }

void MesoHipPairDPDTableForce::read_restart_settings(FILE *fp)
{
  read_gs_settings(fp, comm->me, world, cut_global, seed, mix_flag);
  MESO(meso_pair_dpd_settings(MesoHipContext::get(lmp), style_id, cut_global, seed));
}

void MesoHipPairDPDTableForce::settings(int narg, char **arg)
{
  if (narg != 3) error->all(FLERR, "dpd/tableforce/meso command require: cut_global seed table_length");
  cut_global = atof(arg[0]);
  seed = atoi(arg[1]);
  table_length = atoi(arg[2]);
  MESO(meso_pair_dpd_settings(MesoHipContext::get(lmp), style_id, cut_global, seed));
}

void MesoHipPairDPDTableForce::coeff(int narg, char **arg)
{
  if (narg != 5 && narg != 4 + table_length)
    error->all(FLERR, "Incorrect args for pair dpd/tableforce/meso: type1 type2 gamma sigma < fc_file_name | fc_table >");
  int n = atom->ntypes;
  if (!allocated) allocate_gs();
  double *t = new double[table_length];
  if (narg == 5) {   /* every rank reads the (small) file itself */
    FILE *fp = fopen(arg[4], "r");
    if (!fp) error->one(FLERR, "Cannot open force table file for dpd/tableforce/meso");
    for (int k = 0; k < table_length; k++)
      if (fscanf(fp, "%lf", &t[k]) != 1) error->one(FLERR, "Insufficient parameters in force table file for dpd/tableforce/meso");
    fclose(fp);
  } else {
    for (int k = 0; k < table_length; k++) t[k] = atof(arg[4 + k]);
  }
  int ilo, ihi, jlo, jhi;
  type_bounds(lmp, arg[0], n, ilo, ihi);
  type_bounds(lmp, arg[1], n, jlo, jhi);
  int count = 0;
  for (int i = ilo; i <= ihi; i++)
    for (int j = MAX(jlo, i); j <= jhi; j++) {
      MESO(meso_pair_dpd_tableforce_coeff(MesoHipContext::get(lmp), i, j, atof(arg[2]), atof(arg[3]), table_length, t));
      gamma_f[i][j] = (float) atof(arg[2]); sigma_f[i][j] = (float) atof(arg[3]); cut_f[i][j] = (float) cut_global;
      cut[i][j] = cut_global;
      setflag[i][j] = 1;
      count++;
    }
  delete [] t;
  if (count == 0) error->all(FLERR, "Incorrect args for pair coefficients");
}

void MesoHipPairDPD::settings(int narg, char **arg)
{
  if (narg != 2) error->all(FLERR, "Illegal pair_style command");
  cut_global = atof(arg[0]);
  seed = atoi(arg[1]);
  MESO(meso_pair_dpd_settings(MesoHipContext::get(lmp), style_id, cut_global, seed));
}

void MesoHipPairDPD::allocate()
{
  allocated = 1;
  int n = atom->ntypes;
  memory->create(setflag, n + 1, n + 1, "pair:setflag");
  memory->create(cutsq, n + 1, n + 1, "pair:cutsq");
  memory->create(cut, n + 1, n + 1, "pair:cut");
  memory->create(a0, n + 1, n + 1, "pair:a0");
  memory->create(gamma, n + 1, n + 1, "pair:gamma");
  memory->create(sigma, n + 1, n + 1, "pair:sigma");
  memory->create(expw, n + 1, n + 1, "pair:expw");
  for (int i = 1; i <= n; i++)
    for (int j = i; j <= n; j++) setflag[i][j] = 0;
  MESO(meso_set_mass(MesoHipContext::get(lmp), n, atom->mass));
}

void MesoHipPairDPD::coeff(int narg, char **arg)
{
  if (narg < 6 || narg > 7) error->all(FLERR, "Incorrect args for pair coefficients");
  int n = atom->ntypes;
  if (!allocated) allocate();
  int ilo, ihi, jlo, jhi;
  type_bounds(lmp, arg[0], n, ilo, ihi);
  type_bounds(lmp, arg[1], n, jlo, jhi);
  double cut_one = narg == 7 ? atof(arg[6]) : cut_global;
  int count = 0;
  for (int i = ilo; i <= ihi; i++)
    for (int j = MAX(jlo, i); j <= jhi; j++) {
      MESO(meso_pair_dpd_coeff(MesoHipContext::get(lmp), i, j, atof(arg[2]), atof(arg[3]), atof(arg[4]), atof(arg[5]), cut_one));
      a0[i][j] = atof(arg[2]); gamma[i][j] = atof(arg[3]); sigma[i][j] = atof(arg[4]); expw[i][j] = atof(arg[5]);
      cut[i][j] = cut_one;
      setflag[i][j] = 1;
      count++;
    }
  if (count == 0) error->all(FLERR, "Incorrect args for pair coefficients");
}

/* MesoPairDPD::write_restart (pair_dpd_meso.cu:363-380): per i <= j: setflag, then a0 gamma sigma expw cut */
void MesoHipPairDPD::write_restart(FILE *fp)
{
  write_restart_settings(fp);
  for (int i = 1; i <= atom->ntypes; i++)
    for (int j = i; j <= atom->ntypes; j++) {
      fwrite(&setflag[i][j], sizeof(int), 1, fp);
      if (setflag[i][j]) {
        fwrite(&a0[i][j], sizeof(double), 1, fp);
        fwrite(&gamma[i][j], sizeof(double), 1, fp);
        fwrite(&sigma[i][j], sizeof(double), 1, fp);
        fwrite(&expw[i][j], sizeof(double), 1, fp);
        fwrite(&cut[i][j], sizeof(double), 1, fp);
      }
    }
}

/* MesoPairDPD::read_restart (:386-415): proc 0 reads, everybody gets the values and hands them to the library */
void MesoHipPairDPD::read_restart(FILE *fp)
{
  read_restart_settings(fp);
  if (!allocated) allocate();
  int me = comm->me;
  for (int i = 1; i <= atom->ntypes; i++)
    for (int j = i; j <= atom->ntypes; j++) {
      if (me == 0) fread(&setflag[i][j], sizeof(int), 1, fp);
      MPI_Bcast(&setflag[i][j], 1, MPI_INT, 0, world);
      if (setflag[i][j]) {
        double v[5];
        if (me == 0) fread(v, sizeof(double), 5, fp);
        MPI_Bcast(v, 5, MPI_DOUBLE, 0, world);
        a0[i][j] = v[0]; gamma[i][j] = v[1]; sigma[i][j] = v[2]; expw[i][j] = v[3]; cut[i][j] = v[4];
        MESO(meso_pair_dpd_coeff(MesoHipContext::get(lmp), i, j, v[0], v[1], v[2], v[3], v[4]));
      }
    }
}

/* :421-426 / :432-446: cut_global, seed, mix_flag */
void MesoHipPairDPD::write_restart_settings(FILE *fp)
{
  fwrite(&cut_global, sizeof(double), 1, fp);
  fwrite(&seed, sizeof(int), 1, fp);
  fwrite(&mix_flag, sizeof(int), 1, fp);
}

void MesoHipPairDPD::read_restart_settings(FILE *fp)
{
  if (comm->me == 0) {
    fread(&cut_global, sizeof(double), 1, fp);
    fread(&seed, sizeof(int), 1, fp);
    fread(&mix_flag, sizeof(int), 1, fp);
  }
  MPI_Bcast(&cut_global, 1, MPI_DOUBLE, 0, world);
  MPI_Bcast(&seed, 1, MPI_INT, 0, world);
  MPI_Bcast(&mix_flag, 1, MPI_INT, 0, world);
  MESO(meso_pair_dpd_settings(MesoHipContext::get(lmp), style_id, cut_global, seed));
}

void MesoHipPairDPD::init_style() {}   /* the neighbour table is built inside the library (meso_reneighbor) */

double MesoHipPairDPD::init_one(int i, int j)
{
  if (setflag[i][j] == 0) error->all(FLERR, "All pair coeffs are not set");
  cut[j][i] = cut[i][j];
  return cut[i][j];
}

void MesoHipPairDPD::compute(int eflag, int vflag) { MESO(meso_pair_compute(MesoHipContext::get(lmp), MESO_RANGE_LOCAL, eflag, vflag)); }
void MesoHipPairDPD::compute_bulk(int eflag, int vflag) { MESO(meso_pair_compute(MesoHipContext::get(lmp), MESO_RANGE_BULK, eflag, vflag)); }
void MesoHipPairDPD::compute_border(int eflag, int vflag) { MESO(meso_pair_compute(MesoHipContext::get(lmp), MESO_RANGE_BORDER, eflag, vflag)); }

/* ---------------------------------------------------------------------- bond harmonic/meso */

void MesoHipBondHarmonic::compute(int eflag, int) { MESO(meso_bond_compute(MesoHipContext::get(lmp), eflag)); }

void MesoHipBondHarmonic::allocate()
{
  allocated = 1;
  int n = atom->nbondtypes;
  memory->create(setflag, n + 1, "bond:setflag");
  memory->create(k, n + 1, "bond:k");
  memory->create(r0, n + 1, "bond:r0");
  for (int i = 1; i <= n; i++) setflag[i] = 0;
  MESO(meso_bond_style_harmonic(MesoHipContext::get(lmp), n));
}

void MesoHipBondHarmonic::coeff(int narg, char **arg)
{
  if (narg != 3) error->all(FLERR, "Incorrect args for bond coefficients");
  if (!allocated) allocate();
  int ilo, ihi;
  type_bounds(lmp, arg[0], atom->nbondtypes, ilo, ihi);
  for (int i = ilo; i <= ihi; i++) {
    k[i] = atof(arg[1]); r0[i] = atof(arg[2]);
    MESO(meso_bond_coeff(MesoHipContext::get(lmp), i, k[i], r0[i]));
    setflag[i] = 1;
  }
}

void MesoHipBondHarmonic::write_restart(FILE *fp)
{
  fwrite(&k[1], sizeof(double), atom->nbondtypes, fp);
  fwrite(&r0[1], sizeof(double), atom->nbondtypes, fp);
}

/* BondHarmonic::read_restart src/MOLECULE/bond_harmonic.cpp:163-177, then the coefficients go to the library */
void MesoHipBondHarmonic::read_restart(FILE *fp)
{
  if (!allocated) allocate();
  int n = atom->nbondtypes;
  if (comm->me == 0) {
    fread(&k[1], sizeof(double), n, fp);
    fread(&r0[1], sizeof(double), n, fp);
  }
  MPI_Bcast(&k[1], n, MPI_DOUBLE, 0, world);
  MPI_Bcast(&r0[1], n, MPI_DOUBLE, 0, world);
  for (int i = 1; i <= n; i++) {
    MESO(meso_bond_coeff(MesoHipContext::get(lmp), i, k[i], r0[i]));
    setflag[i] = 1;
  }
}

void MesoHipBondFENE::compute(int eflag, int) { MESO(meso_bond_compute(MesoHipContext::get(lmp), eflag)); }

void MesoHipBondFENE::allocate()
{
  allocated = 1;
  int n = atom->nbondtypes;
  memory->create(setflag, n + 1, "bond:setflag");
  memory->create(k, n + 1, "bond:k");
  memory->create(r0, n + 1, "bond:r0");
  memory->create(epsilon, n + 1, "bond:epsilon");
  memory->create(sigma, n + 1, "bond:sigma");
  for (int i = 1; i <= n; i++) setflag[i] = 0;
  MESO(meso_bond_style_fene(MesoHipContext::get(lmp), n));
}

void MesoHipBondFENE::coeff(int narg, char **arg)
{
  if (narg != 5) error->all(FLERR, "Incorrect args for bond coefficients");
  if (!allocated) allocate();
  int ilo, ihi;
  type_bounds(lmp, arg[0], atom->nbondtypes, ilo, ihi);
  for (int i = ilo; i <= ihi; i++) {
    k[i] = atof(arg[1]); r0[i] = atof(arg[2]); epsilon[i] = atof(arg[3]); sigma[i] = atof(arg[4]);
    MESO(meso_bond_coeff_fene(MesoHipContext::get(lmp), i, k[i], r0[i], epsilon[i], sigma[i]));
    setflag[i] = 1;
  }
}

void MesoHipBondFENE::write_restart(FILE *fp)
{
  fwrite(&k[1], sizeof(double), atom->nbondtypes, fp);
  fwrite(&r0[1], sizeof(double), atom->nbondtypes, fp);
  fwrite(&epsilon[1], sizeof(double), atom->nbondtypes, fp);
  fwrite(&sigma[1], sizeof(double), atom->nbondtypes, fp);
}

/* BondFENE::read_restart src/MOLECULE/bond_fene.cpp:214-232 */
void MesoHipBondFENE::read_restart(FILE *fp)
{
  if (!allocated) allocate();
  int n = atom->nbondtypes;
  if (comm->me == 0) {
    fread(&k[1], sizeof(double), n, fp);
    fread(&r0[1], sizeof(double), n, fp);
    fread(&epsilon[1], sizeof(double), n, fp);
    fread(&sigma[1], sizeof(double), n, fp);
  }
  MPI_Bcast(&k[1], n, MPI_DOUBLE, 0, world);
  MPI_Bcast(&r0[1], n, MPI_DOUBLE, 0, world);
  MPI_Bcast(&epsilon[1], n, MPI_DOUBLE, 0, world);
  MPI_Bcast(&sigma[1], n, MPI_DOUBLE, 0, world);
  for (int i = 1; i <= n; i++) {
    MESO(meso_bond_coeff_fene(MesoHipContext::get(lmp), i, k[i], r0[i], epsilon[i], sigma[i]));
    setflag[i] = 1;
  }
}

/* ---------------------------------------------------------------------- angle_style harmonic/meso */

void MesoHipAngleHarmonic::compute(int eflag, int) { MESO(meso_angle_compute(MesoHipContext::get(lmp), eflag)); }

void MesoHipAngleHarmonic::allocate()
{
  allocated = 1;
  int n = atom->nangletypes;
  memory->create(setflag, n + 1, "angle:setflag");
  memory->create(k, n + 1, "angle:k");
  memory->create(theta0, n + 1, "angle:theta0");
  for (int i = 1; i <= n; i++) setflag[i] = 0;
  MESO(meso_angle_style_harmonic(MesoHipContext::get(lmp), n));
}

static const double MESO_GLUE_PI = 3.14159265358979323846;

void MesoHipAngleHarmonic::coeff(int narg, char **arg)
{
  if (narg != 3) error->all(FLERR, "Incorrect args for angle coefficients");
  if (!allocated) allocate();
  int ilo, ihi;
  type_bounds(lmp, arg[0], atom->nangletypes, ilo, ihi);
  for (int i = ilo; i <= ihi; i++) {
    k[i] = atof(arg[1]);
    theta0[i] = atof(arg[2]) / 180.0 * MESO_GLUE_PI;     /* stored in radians like AngleHarmonic::coeff :157-181 */
    MESO(meso_angle_coeff(MesoHipContext::get(lmp), i, k[i], atof(arg[2])));
    setflag[i] = 1;
  }
}

void MesoHipAngleHarmonic::write_restart(FILE *fp)
{
  fwrite(&k[1], sizeof(double), atom->nangletypes, fp);
  fwrite(&theta0[1], sizeof(double), atom->nangletypes, fp);
}

/* AngleHarmonic::read_restart src/MOLECULE/angle_harmonic.cpp:210-224 (theta0 in radians in the file) */
void MesoHipAngleHarmonic::read_restart(FILE *fp)
{
  if (!allocated) allocate();
  int n = atom->nangletypes;
  if (comm->me == 0) {
    fread(&k[1], sizeof(double), n, fp);
    fread(&theta0[1], sizeof(double), n, fp);
  }
  MPI_Bcast(&k[1], n, MPI_DOUBLE, 0, world);
  MPI_Bcast(&theta0[1], n, MPI_DOUBLE, 0, world);
  for (int i = 1; i <= n; i++) {
    MESO(meso_angle_coeff(MesoHipContext::get(lmp), i, k[i], theta0[i] * 180.0 / MESO_GLUE_PI));
    setflag[i] = 1;
  }
}

/* ---------------------------------------------------------------------- fix nve/meso */

MesoHipFixNVE::MesoHipFixNVE(LAMMPS *lmp, int narg, char **arg) : Fix(lmp, narg, arg)
{
  if (narg < 3) error->all(FLERR, "Illegal fix nve/meso command");
  time_integrate = 1;
}

int MesoHipFixNVE::setmask() { return FixConst::INITIAL_INTEGRATE | FixConst::FINAL_INTEGRATE; }
void MesoHipFixNVE::initial_integrate(int) { MESO(meso_nve_initial(MesoHipContext::get(lmp))); }
void MesoHipFixNVE::final_integrate() { MESO(meso_nve_final(MesoHipContext::get(lmp))); }
void MesoHipFixNVE::reset_dt() { MESO(meso_timestep(MesoHipContext::get(lmp), update->dt)); }

/* ---------------------------------------------------------------------- compute temp/meso */

MesoHipComputeTemp::MesoHipComputeTemp(LAMMPS *lmp, int narg, char **arg) : Compute(lmp, narg, arg)
{
  if (narg != 3) error->all(FLERR, "Illegal compute temp/meso command");
  scalar_flag = 1;
  extscalar = 0;
  tempflag = 1;
}

double MesoHipComputeTemp::compute_scalar()
{
  invoked_scalar = update->ntimestep;
  MESO(meso_compute_temp(MesoHipContext::get(lmp), &scalar));
  return scalar;
}

/* ---------------------------------------------------------------------- compute pe/meso */

MesoHipComputePE::MesoHipComputePE(LAMMPS *lmp, int narg, char **arg) : Compute(lmp, narg, arg)
{
  if (narg < 3) error->all(FLERR, "Illegal compute pe command");
  if (igroup) error->all(FLERR, "Compute pe must use group all");
  scalar_flag = 1;
  extscalar = 1;
  peflag = 1;
  timeflag = 1;
  if (narg == 3) pairflag = bondflag = angleflag = 1;
  else {
    pairflag = bondflag = angleflag = 0;
    for (int iarg = 3; iarg < narg; iarg++) {
      if (strcmp(arg[iarg], "pair") == 0) pairflag = 1;
      else if (strcmp(arg[iarg], "bond") == 0) bondflag = 1;
      else if (strcmp(arg[iarg], "angle") == 0) angleflag = 1;
      else if (strcmp(arg[iarg], "dihedral") == 0 || strcmp(arg[iarg], "improper") == 0) continue;   /* no such device styles */
      else error->all(FLERR, "Illegal compute pe command");
    }
  }
}

double MesoHipComputePE::compute_scalar()
{
  invoked_scalar = update->ntimestep;
  if (update->eflag_global != invoked_scalar) error->all(FLERR, "Energy was not tallied on needed timestep");
  meso_ctx *c = MesoHipContext::get(lmp);
  double e = 0.0;
  scalar = 0.0;                          /* each library call returns the sum over all ranks */
  if (pairflag && force->pair) {
    MESO(meso_compute_pe(c, &e));
    force->pair->eng_vdwl = e;
    scalar += e;
  }
  if (atom->molecular && bondflag && force->bond) {
    MESO(meso_compute_ebond(c, &e));
    force->bond->energy = e;
    scalar += e;
  }
  if (atom->molecular && angleflag && force->angle) {
    MESO(meso_compute_eangle(c, &e));
    force->angle->energy = e;
    scalar += e;
  }
  return scalar;
}

/* ---------------------------------------------------------------------- run_style mvv/meso */

MesoHipIntegrate::MesoHipIntegrate(LAMMPS *lmp, int narg, char **arg) : Integrate(lmp, narg, arg) {}

void MesoHipIntegrate::init()
{
  Integrate::init();
  force->newton = force->newton_pair = force->newton_bond = 0;   /* mvv_meso.cu:101-110 */
  comm->ghost_velocity = 1;
  if (domain->triclinic) error->one(FLERR, "<MESO> triclinic domain not supported in USER-MESO");
  if (force->kspace) error->one(FLERR, "<MESO> kspace not supported in USER-MESO");
}

/* several ranks: MPI_Allgatherv of `nlist` parallel int arrays of local length m; returns the global length and replaces the
   pointers by newly allocated global arrays (one rank: untouched) */
static int gather_lists(LAMMPS *lmp, int m, int nlist, int **lists)
{
  int np = lmp->comm->nprocs;
  if (np == 1) return m;
  int *cnt = new int[np], *dsp = new int[np];
  MPI_Allgather(&m, 1, MPI_INT, cnt, 1, MPI_INT, lmp->world);
  int total = 0;
  for (int p = 0; p < np; p++) { dsp[p] = total; total += cnt[p]; }
  for (int l = 0; l < nlist; l++) {
    int *g = new int[total + 1];
    MPI_Allgatherv(lists[l], m, MPI_INT, g, cnt, dsp, MPI_INT, lmp->world);
    lists[l] = g;
  }
  delete [] cnt; delete [] dsp;
  return total;
}

void MesoHipIntegrate::upload()
{
  meso_ctx *c = MesoHipContext::get(lmp);
  MESO(meso_set_box(c, domain->boxlo, domain->boxhi, domain->periodicity));
  MESO(meso_neighbor(c, neighbor->skin, neighbor->every, neighbor->delay, neighbor->dist_check));
  MESO(meso_timestep(c, update->dt));
  /* each rank hands over the atoms it owns; x[0]/v[0] are the contiguous double[n][3] blocks LAMMPS allocates */
  MESO(meso_atoms_upload(c, atom->nlocal, atom->x[0], atom->v[0], atom->tag, atom->type, atom->mask, atom->image));
  {
    /* every atom must have been accepted by exactly one rank's library context (same sub-domain bounds on both sides) */
    int kept = 0;
    MESO(meso_atoms_count(c, &kept, NULL, NULL));
    bigint mine = kept, all = 0;
    MPI_Allreduce(&mine, &all, 1, MPI_LMP_BIGINT, MPI_SUM, world);
    if (all != atom->natoms) error->all(FLERR, "<MESO> atoms lost while handing the sub-domains to the device");
  }
  if (atom->molecular && atom->bond_per_atom > 0) {
    /* flatten the per-atom bond lists (each bond once: tag < partner when newton_bond is off both atoms store it) */
    int nb = 0;
    for (int i = 0; i < atom->nlocal; i++) nb += atom->num_bond[i];
    int *ti = new int[nb + 1], *tj = new int[nb + 1], *bt = new int[nb + 1];
    int m = 0;
    for (int i = 0; i < atom->nlocal; i++)
      for (int b = 0; b < atom->num_bond[i]; b++)
        if (atom->tag[i] < atom->bond_atom[i][b] || force->newton_bond) {
          ti[m] = atom->tag[i]; tj[m] = atom->bond_atom[i][b]; bt[m] = atom->bond_type[i][b]; m++;
        }
    /* the library wants the whole Bonds section on every rank (it keeps what belongs to its atoms) */
    MESO(meso_special_bonds(c, force->special_lj[1], force->special_lj[2], force->special_lj[3]));
    int *all[3] = {ti, tj, bt};
    int mall = gather_lists(lmp, m, 3, all);
    MESO(meso_bonds_upload(c, mall, all[0], all[1], all[2]));
    if (all[0] != ti) { delete [] all[0]; delete [] all[1]; delete [] all[2]; }
    delete [] ti; delete [] tj; delete [] bt;
    if (atom->angle_per_atom > 0) {
      /* each angle once: the copy stored on its apex atom (with newton_bond off all three atoms store it) */
      int na = 0;
      for (int i = 0; i < atom->nlocal; i++) na += atom->num_angle[i];
      int *a1 = new int[na + 1], *a2 = new int[na + 1], *a3 = new int[na + 1], *at = new int[na + 1];
      int q = 0;
      for (int i = 0; i < atom->nlocal; i++)
        for (int a = 0; a < atom->num_angle[i]; a++)
          if (atom->tag[i] == atom->angle_atom2[i][a]) {
            a1[q] = atom->angle_atom1[i][a]; a2[q] = atom->angle_atom2[i][a]; a3[q] = atom->angle_atom3[i][a];
            at[q] = atom->angle_type[i][a]; q++;
          }
      int *alla[4] = {a1, a2, a3, at};
      int qall = gather_lists(lmp, q, 4, alla);
      MESO(meso_angles_upload(c, qall, alla[0], alla[1], alla[2], alla[3]));
      if (alla[0] != a1) { delete [] alla[0]; delete [] alla[1]; delete [] alla[2]; delete [] alla[3]; }
      delete [] a1; delete [] a2; delete [] a3; delete [] at;
    }
  }
}

void MesoHipIntegrate::download()
{
  meso_ctx *c = MesoHipContext::get(lmp);
  int n = 0;
  MESO(meso_atoms_count(c, &n, NULL, NULL));
  if (n > atom->nmax) atom->avec->grow(n);
  atom->nlocal = n;
  MESO(meso_atoms_download(c, atom->x[0], atom->v[0], atom->f[0], atom->tag, atom->type, atom->image));
}

void MesoHipIntegrate::setup()
{
  update->setupflag = 1;
  upload();
  MESO(meso_step_advance(MesoHipContext::get(lmp), update->ntimestep));
  MESO(meso_setup(MesoHipContext::get(lmp)));
  download();
  output->setup();
  update->setupflag = 0;
}

void MesoHipIntegrate::setup_minimal(int) { setup(); }

void MesoHipIntegrate::run(int n)
{
  meso_ctx *c = MesoHipContext::get(lmp);
  int done = 0;
  while (done < n) {
    bigint next = output->next;
    int chunk = n - done;
    if (next > update->ntimestep && next - update->ntimestep < chunk) chunk = (int) (next - update->ntimestep);
    MESO(meso_run(c, chunk));
    update->ntimestep += chunk;
    done += chunk;
    if (update->ntimestep == output->next) {
      download();                                  /* transfer_pre_output, atom_meso.cu:258-266 */
      output->write(update->ntimestep);
    }
  }
  download();
}

void MesoHipIntegrate::cleanup() {}
