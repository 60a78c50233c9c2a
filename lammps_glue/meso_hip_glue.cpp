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
#include "neighbor.h"
#include "output.h"
#include "universe.h"
#include "update.h"

using namespace LAMMPS_NS;

#define MESO(call) MesoHipContext::check(lmp, (call), FLERR)

static meso_ctx *g_ctx = NULL;

meso_ctx *MesoHipContext::get(LAMMPS *lmp)
{
  if (!g_ctx) {
    /* one GPU per rank on the node, like the -device flag of the reference */
    int rc = meso_init(-(lmp->universe->me), &g_ctx);
    check(lmp, rc, FLERR);
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

MesoHipPairDPDPolyForce::MesoHipPairDPDPolyForce(LAMMPS *lmp) : MesoHipPairDPD(lmp) { style_id = MESO_PAIR_DPD_POLYFORCE; }

void MesoHipPairDPDPolyForce::coeff(int narg, char **arg)
{
  if (narg < 6 || narg != 6 + atoi(arg[4])) error->all(FLERR, "Incorrect args for pair coefficients");
  int n = atom->ntypes;
  if (!allocated) {
    allocated = 1;
    memory->create(setflag, n + 1, n + 1, "pair:setflag");
    memory->create(cutsq, n + 1, n + 1, "pair:cutsq");
    memory->create(cut, n + 1, n + 1, "pair:cut");
    for (int i = 1; i <= n; i++)
      for (int j = i; j <= n; j++) setflag[i][j] = 0;
    MESO(meso_set_mass(MesoHipContext::get(lmp), n, atom->mass));
  }
  int ilo, ihi, jlo, jhi;
  force->bounds(arg[0], n, ilo, ihi);
  force->bounds(arg[1], n, jlo, jhi);
  int order = atoi(arg[4]);
  double *c = new double[order + 1];
  for (int k = 0; k <= order; k++) c[k] = atof(arg[5 + k]);
  int count = 0;
  for (int i = ilo; i <= ihi; i++)
    for (int j = MAX(jlo, i); j <= jhi; j++) {
      MESO(meso_pair_dpd_polyforce_coeff(MesoHipContext::get(lmp), i, j, atof(arg[2]), atof(arg[3]), order, c));
      cut[i][j] = cut_global;
      setflag[i][j] = 1;
      count++;
    }
  delete [] c;
  if (count == 0) error->all(FLERR, "Incorrect args for pair coefficients");
}

MesoHipPairDPDTableForce::MesoHipPairDPDTableForce(LAMMPS *lmp) : MesoHipPairDPD(lmp), table_length(0) { style_id = MESO_PAIR_DPD_TABLEFORCE; }

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
  if (!allocated) {
    allocated = 1;
    memory->create(setflag, n + 1, n + 1, "pair:setflag");
    memory->create(cutsq, n + 1, n + 1, "pair:cutsq");
    memory->create(cut, n + 1, n + 1, "pair:cut");
    for (int i = 1; i <= n; i++)
      for (int j = i; j <= n; j++) setflag[i][j] = 0;
    MESO(meso_set_mass(MesoHipContext::get(lmp), n, atom->mass));
  }
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
  force->bounds(arg[0], n, ilo, ihi);
  force->bounds(arg[1], n, jlo, jhi);
  int count = 0;
  for (int i = ilo; i <= ihi; i++)
    for (int j = MAX(jlo, i); j <= jhi; j++) {
      MESO(meso_pair_dpd_tableforce_coeff(MesoHipContext::get(lmp), i, j, atof(arg[2]), atof(arg[3]), table_length, t));
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
  force->bounds(arg[0], n, ilo, ihi);
  force->bounds(arg[1], n, jlo, jhi);
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

void MesoHipBondHarmonic::coeff(int narg, char **arg)
{
  if (narg != 3) error->all(FLERR, "Incorrect args for bond coefficients");
  if (!allocated) {
    allocated = 1;
    memory->create(setflag, atom->nbondtypes + 1, "bond:setflag");
    for (int i = 1; i <= atom->nbondtypes; i++) setflag[i] = 0;
    MESO(meso_bond_style_harmonic(MesoHipContext::get(lmp), atom->nbondtypes));
  }
  int ilo, ihi;
  force->bounds(arg[0], atom->nbondtypes, ilo, ihi);
  for (int i = ilo; i <= ihi; i++) {
    MESO(meso_bond_coeff(MesoHipContext::get(lmp), i, atof(arg[1]), atof(arg[2])));
    setflag[i] = 1;
  }
}

void MesoHipBondFENE::compute(int eflag, int) { MESO(meso_bond_compute(MesoHipContext::get(lmp), eflag)); }

void MesoHipBondFENE::coeff(int narg, char **arg)
{
  if (narg != 5) error->all(FLERR, "Incorrect args for bond coefficients");
  if (!allocated) {
    allocated = 1;
    memory->create(setflag, atom->nbondtypes + 1, "bond:setflag");
    for (int i = 1; i <= atom->nbondtypes; i++) setflag[i] = 0;
    MESO(meso_bond_style_fene(MesoHipContext::get(lmp), atom->nbondtypes));
  }
  int ilo, ihi;
  force->bounds(arg[0], atom->nbondtypes, ilo, ihi);
  for (int i = ilo; i <= ihi; i++) {
    MESO(meso_bond_coeff_fene(MesoHipContext::get(lmp), i, atof(arg[1]), atof(arg[2]), atof(arg[3]), atof(arg[4])));
    setflag[i] = 1;
  }
}

/* ---------------------------------------------------------------------- angle_style harmonic/meso */

void MesoHipAngleHarmonic::compute(int eflag, int) { MESO(meso_angle_compute(MesoHipContext::get(lmp), eflag)); }

void MesoHipAngleHarmonic::coeff(int narg, char **arg)
{
  if (narg != 3) error->all(FLERR, "Incorrect args for angle coefficients");
  if (!allocated) {
    allocated = 1;
    memory->create(setflag, atom->nangletypes + 1, "angle:setflag");
    for (int i = 1; i <= atom->nangletypes; i++) setflag[i] = 0;
    MESO(meso_angle_style_harmonic(MesoHipContext::get(lmp), atom->nangletypes));
  }
  int ilo, ihi;
  force->bounds(arg[0], atom->nangletypes, ilo, ihi);
  for (int i = ilo; i <= ihi; i++) {
    MESO(meso_angle_coeff(MesoHipContext::get(lmp), i, atof(arg[1]), atof(arg[2])));
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

void MesoHipIntegrate::upload()
{
  meso_ctx *c = MesoHipContext::get(lmp);
  MESO(meso_set_box(c, domain->boxlo, domain->boxhi, domain->periodicity));
  MESO(meso_neighbor(c, neighbor->skin, neighbor->every, neighbor->delay, neighbor->dist_check));
  MESO(meso_timestep(c, update->dt));
  /* each rank hands over the atoms it owns; x[0]/v[0] are the contiguous double[n][3] blocks LAMMPS allocates */
  MESO(meso_atoms_upload(c, atom->nlocal, atom->x[0], atom->v[0], atom->tag, atom->type, atom->mask, atom->image));
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
    /* one rank: the list is complete; several ranks: MPI_Allgatherv the three arrays before this call */
    MESO(meso_special_bonds(c, force->special_lj[1], force->special_lj[2], force->special_lj[3]));
    MESO(meso_bonds_upload(c, m, ti, tj, bt));
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
      MESO(meso_angles_upload(c, q, a1, a2, a3, at));
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
