/* LAMMPS-side binding of libmeso_hip.so: the style classes a maintainer adds to a LAMMPS tree (next to, or
   instead of, src/USER-MESO) so that the input decks of example/simple keep working unchanged:

     atom_style dpd/atomic/meso  ->  MesoHipAtomVecDPDAtomic (AtomStyle, replaces AtomVecDPDAtomic atom_vec_dpd_atomic_meso.h:3)
     atom_style dpd/bond/meso    ->  MesoHipAtomVecDPDBond   (AtomStyle, replaces AtomVecDPDBond   atom_vec_dpd_bond_meso.h:3)
     atom_style dpd/angle/meso   ->  MesoHipAtomVecDPDAngle  (AtomStyle, replaces AtomVecDPDAngle  atom_vec_dpd_angle_meso.h:3)
     run_style  mvv/meso         ->  MesoHipIntegrate   (IntegrateStyle, replaces ModifiedVerlet  mvv_meso.h:3-4)
     pair_style dpd/meso         ->  MesoHipPairDPD     (PairStyle,      replaces MesoPairDPD     pair_dpd_meso.h:3)
     pair_style dpd/fast/meso    ->  MesoHipPairDPDFast (PairStyle,      replaces MesoPairDPDFast pair_dpd_fast_meso.h:3)
     fix        nve/meso         ->  MesoHipFixNVE      (FixStyle,       replaces FixNVEMeso      fix_nve_meso.h:3)
     compute    temp/meso        ->  MesoHipComputeTemp (ComputeStyle,   replaces MesoComputeTemp compute_temp_meso.h)
     compute    pe/meso          ->  MesoHipComputePE   (ComputeStyle,   replaces MesoComputePE   compute_pe_meso.h:3)

   The classes hold no HIP code: every virtual forwards to the C ABI of include/meso_hip.h.  The particle state lives
   on the GPU between timesteps; LAMMPS' host arrays are refreshed (meso_atoms_download) only when the host needs
   them (thermo / dump steps, end of run), which is what transfer_pre_output does in the reference
   (atom_meso.cu:258-266).  Registration uses LAMMPS' own macro factory (src/force.cpp:81-86, src/update.cpp:298-318):
   Make.sh style picks the *_CLASS blocks up from this header. */

#ifdef ATOM_CLASS
AtomStyle(dpd/atomic/meso,MesoHipAtomVecDPDAtomic)
AtomStyle(dpd/bond/meso,MesoHipAtomVecDPDBond)
AtomStyle(dpd/angle/meso,MesoHipAtomVecDPDAngle)
#elif defined(PAIR_CLASS)
PairStyle(dpd/meso,MesoHipPairDPD)
PairStyle(dpd/fast/meso,MesoHipPairDPDFast)
PairStyle(dpd/mini/meso,MesoHipPairDPDMini)
PairStyle(dpd/polyforce/meso,MesoHipPairDPDPolyForce)
PairStyle(dpd/tableforce/meso,MesoHipPairDPDTableForce)
#elif defined(BOND_CLASS)
BondStyle(harmonic/meso,MesoHipBondHarmonic)
BondStyle(fene/meso,MesoHipBondFENE)
#elif defined(ANGLE_CLASS)
AngleStyle(harmonic/meso,MesoHipAngleHarmonic)
#elif defined(FIX_CLASS)
FixStyle(nve/meso,MesoHipFixNVE)
#elif defined(COMPUTE_CLASS)
ComputeStyle(temp/meso,MesoHipComputeTemp)
ComputeStyle(pe/meso,MesoHipComputePE)
#elif defined(INTEGRATE_CLASS)
IntegrateStyle(mvv/meso,MesoHipIntegrate)
IntegrateStyle(verlet/meso,MesoHipIntegrate)
#else

#ifndef LMP_MESO_HIP_GLUE_H
#define LMP_MESO_HIP_GLUE_H

#include "angle.h"
#include "atom_vec_angle.h"      /* MOLECULE package (make yes-molecule), as for the reference's dpd/bond|angle/meso styles */
#include "atom_vec_atomic.h"
#include "atom_vec_bond.h"
#include "bond.h"
#include "compute.h"
#include "fix.h"
#include "integrate.h"
#include "pair.h"
#include "meso_hip.h"

namespace LAMMPS_NS {

/* one context per LAMMPS instance (= per MPI rank = per GPU), shared by the styles below */
struct MesoHipContext {
  /* meso_init on first use (device = local rank, src/lammps.cpp:432-452).  With more than one MPI rank the first call
     also binds the library's ghost exchange (replacing MesoComm, comm_meso.cu:41-186,256-550) to LAMMPS' decomposition:
     meso_comm_init(ctx, nprocs, grid rank of comm->myloc, comm->procgrid, RCCL, ncclUniqueId of MPI rank 0 (MPI_Bcast)) */
  static meso_ctx *get(class LAMMPS *lmp);
  static void check(class LAMMPS *lmp, int rc, const char *file, int line);   /* rc != 0 -> error->one(file,line,meso_last_error()) */
};

/* atom styles: the host arrays are LAMMPS' own (read_data, dumps, restarts keep working); between output steps they are
   only a mirror of the device state, which lives in the library.  Pinning / grow_device / the transfer engine of
   MesoAtomVec (atom_vec_meso.cu:77-140,220-323) have no counterpart: meso_atoms_upload/download borrow the arrays. */
class MesoHipAtomVecDPDAtomic : public AtomVecAtomic {
 public:
  MesoHipAtomVecDPDAtomic(class LAMMPS *lmp) : AtomVecAtomic(lmp) { cudable = 1; }
};
class MesoHipAtomVecDPDBond : public AtomVecBond {
 public:
  MesoHipAtomVecDPDBond(class LAMMPS *lmp) : AtomVecBond(lmp) { cudable = 1; }
};
class MesoHipAtomVecDPDAngle : public AtomVecAngle {
 public:
  MesoHipAtomVecDPDAngle(class LAMMPS *lmp) : AtomVecAngle(lmp) { cudable = 1; }
};

class MesoHipPairDPD : public Pair {
 public:
  MesoHipPairDPD(class LAMMPS *);
  virtual ~MesoHipPairDPD() {}
  void compute(int, int);              /* meso_pair_compute(ctx, MESO_RANGE_LOCAL, eflag, vflag) */
  void compute_bulk(int, int);         /* ... MESO_RANGE_BULK   (pair_dpd_meso.cu:241-248) */
  void compute_border(int, int);       /* ... MESO_RANGE_BORDER (pair_dpd_meso.cu:250-257) */
  void settings(int, char **);         /* pair_style dpd/meso rc seed           -> meso_pair_dpd_settings */
  void coeff(int, char **);            /* pair_coeff i j a0 gamma sigma s [rc]  -> meso_pair_dpd_coeff */
  void init_style();
  double init_one(int, int);
  /* restart file records with the byte layout of MesoPairDPD::write_restart / write_restart_settings (pair_dpd_meso.cu:363-447) */
  void write_restart(FILE *);
  void read_restart(FILE *);
  void write_restart_settings(FILE *);
  void read_restart_settings(FILE *);
 protected:
  void allocate();
  int style_id;                        /* MESO_PAIR_DPD or MESO_PAIR_DPD_FAST */
  double cut_global;
  int seed;
  double **cut, **a0, **gamma, **sigma, **expw;     /* host copies for the restart file; the library holds the working tables */
};

class MesoHipPairDPDFast : public MesoHipPairDPD {
 public:
  MesoHipPairDPDFast(class LAMMPS *);
};

/* pair_style dpd/mini/meso <unused> seed; pair_coeff * * a0 gamma sigma (replaces MesoPairDPDMini pair_dpd_minimal_meso.h:3) */
class MesoHipPairDPDMini : public MesoHipPairDPD {
 public:
  MesoHipPairDPDMini(class LAMMPS *);
  void settings(int, char **);
  void coeff(int, char **);
};

/* pair_style dpd/polyforce/meso rc seed; pair_coeff i j gamma sigma order c_order ... c_0 (replaces MesoPairDPDPolyForce
   pair_dpd_polyforce_meso.h:3) */
class MesoHipPairDPDPolyForce : public MesoHipPairDPD {
 public:
  MesoHipPairDPDPolyForce(class LAMMPS *);
  void coeff(int, char **);            /* -> meso_pair_dpd_polyforce_coeff */
  /* records of MesoPairDPDPolyForce::write_restart (pair_dpd_polyforce_meso.cu:370-447): float gamma sigma cut per set pair,
     settings float cut_global, int seed, int mix_flag; like the reference the polynomial itself is NOT in the file:
     read_restart warns and pair_coeff has to be given again */
  void write_restart(FILE *);
  void read_restart(FILE *);
  void write_restart_settings(FILE *);
  void read_restart_settings(FILE *);
 protected:
  void allocate_gs();
  float **gamma_f, **sigma_f, **cut_f;
};

/* pair_style dpd/tableforce/meso rc seed table_length; pair_coeff i j gamma sigma < file | table_length values > (replaces
   MesoPairDPDTableForce pair_dpd_tableforce_meso.h:3) */
class MesoHipPairDPDTableForce : public MesoHipPairDPD {
 public:
  MesoHipPairDPDTableForce(class LAMMPS *);
  void settings(int, char **);
  void coeff(int, char **);            /* -> meso_pair_dpd_tableforce_coeff */
  /* records of MesoPairDPDTableForce::write_restart (pair_dpd_tableforce_meso.cu:394-470), same layout as polyforce; neither
     the table nor its length is in the file (reference behaviour): pair_style and pair_coeff are given again after a restart */
  void write_restart(FILE *);
  void read_restart(FILE *);
  void write_restart_settings(FILE *);
  void read_restart_settings(FILE *);
 protected:
  void allocate_gs();
  int table_length;
  float **gamma_f, **sigma_f, **cut_f;
};

/* bond_style harmonic/meso (replaces MesoBondHarmonic, bond_harmonic_meso.h:3); the Bonds section is handed over once by
   MesoHipIntegrate::upload through meso_bonds_upload (after meso_special_bonds with force->special_lj[1..3]) */
class MesoHipBondHarmonic : public Bond {
 public:
  MesoHipBondHarmonic(class LAMMPS *lmp) : Bond(lmp), k(NULL), r0(NULL) {}
  void compute(int, int);              /* meso_bond_compute */
  void coeff(int, char **);            /* bond_coeff type K r0 -> meso_bond_coeff */
  double equilibrium_distance(int i) { return r0[i]; }
  void write_restart(FILE *);          /* BondHarmonic::write_restart src/MOLECULE/bond_harmonic.cpp:153-157: k[1..n], r0[1..n] */
  void read_restart(FILE *);
  double single(int, double, int, int, double &) { return 0.0; }
 protected:
  void allocate();
  double *k, *r0;
};

/* bond_style fene/meso (replaces MesoBondFENE, bond_fene_meso.h:3): bond_coeff type K R0 epsilon sigma */
class MesoHipBondFENE : public Bond {
 public:
  MesoHipBondFENE(class LAMMPS *lmp) : Bond(lmp), k(NULL), r0(NULL), epsilon(NULL), sigma(NULL) {}
  void compute(int, int);              /* meso_bond_compute */
  void coeff(int, char **);            /* -> meso_bond_coeff_fene */
  double equilibrium_distance(int i) { return 0.97 * sigma[i]; }   /* BondFENE::equilibrium_distance */
  void write_restart(FILE *);          /* BondFENE::write_restart src/MOLECULE/bond_fene.cpp:202-208: k, r0, epsilon, sigma */
  void read_restart(FILE *);
  double single(int, double, int, int, double &) { return 0.0; }
 protected:
  void allocate();
  double *k, *r0, *epsilon, *sigma;
};

/* angle_style harmonic/meso (replaces MesoAngleHarmonic, angle_harmonic_meso.h:3); the Angles section is handed over by
   MesoHipIntegrate::upload through meso_angles_upload, after the bonds */
class MesoHipAngleHarmonic : public Angle {
 public:
  MesoHipAngleHarmonic(class LAMMPS *lmp) : Angle(lmp), k(NULL), theta0(NULL) {}
  void compute(int, int);              /* meso_angle_compute */
  void coeff(int, char **);            /* angle_coeff type K theta0[degrees] -> meso_angle_coeff */
  double equilibrium_angle(int i) { return theta0[i]; }
  void write_restart(FILE *);          /* AngleHarmonic::write_restart src/MOLECULE/angle_harmonic.cpp:200-204: k, theta0 (radians) */
  void read_restart(FILE *);
  double single(int, int, int, int) { return 0.0; }
 protected:
  void allocate();
  double *k, *theta0;
};

class MesoHipFixNVE : public Fix {
 public:
  MesoHipFixNVE(class LAMMPS *, int, char **);
  int setmask();
  void initial_integrate(int);         /* meso_nve_initial */
  void final_integrate();              /* meso_nve_final   */
  void reset_dt();                     /* meso_timestep    */
};

class MesoHipComputeTemp : public Compute {
 public:
  MesoHipComputeTemp(class LAMMPS *, int, char **);
  void init() {}
  double compute_scalar();             /* meso_compute_temp (global sum done inside the library) */
};

/* compute ID all pe/meso [pair] [bond] [angle] (MesoComputePE::compute_scalar compute_pe_meso.cu:66-125): sums of the per-atom
   energies kept on the device, global sum inside the library */
class MesoHipComputePE : public Compute {
 public:
  MesoHipComputePE(class LAMMPS *, int, char **);
  void init() {}
  double compute_scalar();             /* meso_compute_pe (+ meso_compute_ebond, meso_compute_eangle) */
 private:
  int pairflag, bondflag, angleflag;
};

class MesoHipIntegrate : public Integrate {
 public:
  MesoHipIntegrate(class LAMMPS *, int, char **);
  void init();                         /* forces newton off / ghost velocities on like mvv_meso.cu:79-133 */
  void setup();                        /* upload host arrays (meso_set_box/mass/atoms_upload), meso_setup */
  void setup_minimal(int);
  void run(int);                       /* meso_run in chunks that end on output->next, then download */
  void cleanup();
 protected:
  void upload();
  void download();
};

}

#endif
#endif
