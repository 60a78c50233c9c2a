/* meso_hip.h - C ABI of libmeso_hip.so: the MI355X-native DPD hot path of LAMMPS' USER-MESO package.
 *
 * Drop-in boundary.  Each entry point names the reference interface it replaces
 * (paths relative to /root/reference/src/USER-MESO unless stated).  The library owns all device memory;
 * host pointers are borrowed for the duration of a call only.  One context per GPU / per rank; a context
 * is not thread-safe, different contexts are independent.  Every function returns 0 on success and a
 * non-zero status otherwise; the message is available from meso_last_error() (the LAMMPS glue maps it to
 * error->one(FLERR,msg), replacing raise(SIGABRT) of util_meso.h:96-97 and exit(0) of engine_meso.h:139).
 */
#ifndef MESO_HIP_H
#define MESO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct meso_ctx meso_ctx;

enum { MESO_OK = 0, MESO_ERR_ARG = 1, MESO_ERR_HIP = 2, MESO_ERR_STATE = 3, MESO_ERR_OVERFLOW = 4, MESO_ERR_COMM = 5 };

/* pair styles: PairStyle(dpd/meso,MesoPairDPD) pair_dpd_meso.h:3 ; PairStyle(dpd/fast/meso,...) pair_dpd_fast_meso.h:3 */
/* Units of the fp32 styles (dpd/fast/meso, dpd/mini/meso, dpd/polyforce/meso, dpd/tableforce/meso): the force kernel adds an atom's
 * pair forces as 32-bit fixed point with 16 fractional bits - a term is truncated to 1.5e-5 force units and a component of the sum
 * holds +-32768.  Reduced DPD units stay three orders of magnitude inside that.  A sum beyond HALF the range is reported by meso_run /
 * meso_pair_compute as MESO_ERR_OVERFLOW ("... fixed-point sums ...", tests/test_gpu_parity.py::test_fp32_force_sums_out_of_range_are_
 * reported) instead of wrapping; dpd/meso adds in 64 bits (36 fractional: 1.5e-11 resolution, +-1.3e8) and is the style for such decks. */
/* MESO_PAIR_DPD_MINI: PairStyle(dpd/mini/meso) pair_dpd_minimal_meso.h:3 - fp32 arithmetic, cutoff 1, one (a0, gamma, sigma)
 * for all types (pair_coeff * * a0 gamma sigma), pair noise from the logistic map mean0var1<8> (:50-89) instead of TEA */
enum { MESO_PAIR_DPD = 0, MESO_PAIR_DPD_FAST = 1, MESO_PAIR_DPD_MINI = 2, MESO_PAIR_DPD_POLYFORCE = 3, MESO_PAIR_DPD_TABLEFORCE = 4 };
/* work ranges, AtomAttribute::LOCAL/BULK/BORDER util_meso.h:43-74, resolve_work_range atom_vec_meso.cu:194-218 */
enum { MESO_RANGE_LOCAL = 0, MESO_RANGE_BULK = 1, MESO_RANGE_BORDER = 2 };
/* ghost transports */
enum { MESO_TRANSPORT_SELF = 0, MESO_TRANSPORT_RCCL = 1, MESO_TRANSPORT_HOST = 2, MESO_TRANSPORT_LOCAL = 3 };

const char *meso_last_error(void);
int meso_version(void);

/* ---- device runtime: MesoDevice::init engine_meso.cu:38-96, LAMMPS -device flag src/lammps.cpp:170-191,432-455 */
int meso_init(int device, meso_ctx **ctx);
int meso_finalize(meso_ctx *ctx);
int meso_device_sync(meso_ctx *ctx);
/* engine options (profiling windows, kernel variants); unknown keys are an error.  Defaults are the measured best:
 *   pair_kernel   2  ring kernel (both styles) | 0 lane per atom (the kernel that also books energy and virial); the tile, MLP
 *                    and brick kernels and the bin-sorted / brick layouts of round 1 were retired (option layout accepts 2 only)
 *   neigh_kernel  1  wave-per-bin tile builder | 0 lane-per-atom cell builder (also the fallback for very wide rows)
 *   fuse_step     1  final(s) + initial(s+1) + merge(s+1) in one pass between the steps of one run
 *   fuse_pair     1  ... inside the ring kernel's epilogue (forces are then not stored on those steps)
 *   pair_share    1  ring kernel: pairs inside one aligned 256-atom group are evaluated once (Newton pairing)
 *   tile_plan     0  the tile builder computes each brick's plan (halo runs, slot map) itself | 1 separate k_brick_plan launch (what rows
 *                    shorter than 64 entries use by themselves: test_short_cutoff_rows_of_32_survive_several_fused_rebuilds)
 *   reorder_sort  0  locals reordered by counting per [border][Morton(bin)] code | 1 by rocPRIM's radix sort (north_star's wording; same
 *                    order: test_reorder_by_counting_equals_reorder_by_sorting)
 *   ghost_sort    0  ghosts binned by counting per Morton code | 1 by sorting them (same storage order, same test)
 *   pair_npart    0  ring kernel: lanes per atom; 0 = by launch size (2 up to 163 840 atoms, else 1) | 1 | 2 | 4
 *   fuse_clear    1  force kernel writes f instead of clear + accumulate
 *   fuse_bonds    1  systems with bonds and no angles: the bonds of an atom are evaluated in the force kernel's step-boundary
 *                    epilogue (no bond kernel launch between rebuild-free steps); 0 = bond kernel first, epilogue adds its forces
 *   overlap       1  several ranks: ghost refresh on a side stream under the bulk force kernel
 *   brick_margin  1  multiplier on the expected brick-neighbourhood population (the LDS stage of the tile list builder; the engine
 *                    also grows it by itself: a first build that overflows is repeated, later ones grow from the high-water mark)
 *   async_counts  1  one rank: a rebuild does not wait for the host - launch bounds come from the previous rebuild's counts,
 *                    kernels mask with the device-side counts, the host reads them (pinned memory) when it next needs them;
 *                    several ranks: the border messages of a rebuild have a fixed capacity derived from the previous
 *                    rebuild's counts (count * (1 + mr_cap_margin) + 256, the same number on both sides) and carry their counts
 *                    in a header, so the ghost stage needs neither a count exchange nor a host round trip
 *   mig_cap_floor 64  several ranks, async_counts: a migration message has the capacity 2 * (count of the previous rebuild) + floor and
 *                    carries its counts in a header (no separate count exchange); one that does not fit is sent again, exactly
 *                    One rank: a ghost list / border range / cell bucket that outgrows what the previous rebuild reserved is NOT an error -
 *                    every launch behind that rebuild stores nothing (a device flag) and meso_run redoes the rebuild through the
 *                    synchronous path with regrown capacities (test_an_outgrown_capacity_is_redone_not_fatal; MesoComm::borders regrows
 *                    its buffers on the fly as well, comm_meso.cu:122,138,179-181).
 *   mr_cap_margin 0.5  see async_counts (several ranks: a message that outgrows its capacity is an error, reported at the end of run(); between
 *                    two rebuilds a slab's ghosts can grow by at most the atoms of a layer as thick as the largest displacement
 *                    next to it - 11.5 % at equal density with the default skin)
 *   row_part     -1  1 = the list builder writes every row in two sections (meso_neigh_parts): what the atom evaluates in front,
 *                    mirrored in-group entries behind; the ring kernel walks the front section only.  0 = plain rows, pairing
 *                    decided per entry and step from the two indices (round-4 form; bit-identical forces).  -1 = two sections when
 *                    they pay: rebuild interval of at least 4 steps (fp32 styles) / 2 steps (dpd/meso), or neigh_modify check yes
 *   split_gather -1  one rank: the rebuild's placing kernel only orders and a streaming pass moves the payload
 *                    (-1 = boxes of at least 50 000 local atoms (25^3: +2-3 %, 32^3: the same, larger: faster), 0 off, 1 on)
 *   ghost_epilogue -1  one rank: the step-boundary epilogue of the force kernel also writes the merged pairs of each atom's periodic
 *                    images (no k_pack_forward launch between rebuilds); -1 = on (without xcd_balance: up to 524 288 local atoms), 0 off, 1 on
 *   overlap_rebuild 0  with async_counts: 1 = reorder of the locals on the main stream, border lists + ghost creation + ghost
 *                    binning on the side stream, joined by events (north_star's "reorder on a side stream overlapped with halo pack":
 *                    same neighbour sets and forces, measured 4-7 % slower at every size; test_rebuild_variants_give_the_same_trajectory)
 *   fuse_count    1  one rank: on the step in front of a rebuild the force kernel's step-boundary epilogue also runs the rebuild's first
 *                    kernel (wrap, cell code, rank inside the cell, bucket entry: k_fr_count) over the positions it has just written;
 *                    0 = the count as a launch of its own (test_rebuild_variants_give_the_same_trajectory, the 64^3 case in
 *                    test_config2_64cube_two_section_rows_are_bit_identical_to_plain_rows)
 *   merge_ghosts  1  one rank, with split_gather: the ghost tiles of a rebuild run in the launch of its gather (they read the old order
 *                    through the permutation; the count books every border atom's periodic images): one launch less, the ghost
 *                    tiles' latency chains under the stream (test_rebuild_variants_give_the_same_trajectory)
 *   lean_boundary 1  the force kernel's step-boundary epilogue takes the atom's type from the merged coordinate record it holds and the mass
 *                    from the per-type table, and skips the mask of group "all" (16 bytes per atom less to read); 0 = per-atom arrays
 *                    (test_rebuild_variants_give_the_same_trajectory)
 *   xcd_balance   1  force launches of more than one round of workgroups: bulk and border workgroups are dealt out over the eight XCDs
 *                    separately (in Morton order an XCD's border share lies next to its bulk share); 0 = one contiguous range of atoms
 *                    per XCD, which leaves every border atom - a fifth more pairs, the periodic images - to the last XCD
 *                    (test_config2_64cube_two_section_rows_are_bit_identical_to_plain_rows)
 *   report_poll   1  one rank, fused rebuild: the kernel that reports the rebuild's counts into pinned host memory writes the rebuild's
 *                    sequence number behind them and the host polls that word when it next needs the counts; 0 = an event recorded behind
 *                    the rebuild's last launch (a marker packet: ~6 us of bubble in front of the list builder on every rebuild;
 *                    test_rebuild_variants_give_the_same_trajectory runs both)
 *   tile_persist  0  1 = the tile list builder runs as persistent workgroups (as many as the card holds at once) that draw their bricks
 *                    from 32 counters - the reference's builder is a grid-stride loop, neigh_build_meso.cu:58; measured 6 % slower than
 *                    one workgroup per brick (64^3: 219 against 206 us per build), so off (test_config2_64cube_two_section_rows_are_bit_
 *                    identical_to_plain_rows keeps it alive)
 *   Options whose non-default value lost in every measurement for three rounds stay as ONE tested alternative each, not as tuning knobs:
 *   ghost_sort 1 (ghosts binned by sorting: test_reorder_by_counting_equals_reorder_by_sorting), tile_plan 1 (separate plan launch; what rows
 *   shorter than 64 entries use by themselves: test_short_cutoff_rows_of_32_survive_several_fused_rebuilds), and - several ranks -
 *   border_runs 0, mig_slim 0, border_fused 0, refresh_direct 0, refresh_epilogue 0 (the chains of small launches the round-3/4 kernels
 *   replaced: all five in test_borders_without_host_round_trip_equal_the_synchronous_path of tests/test_gpu_multirank.py, the last two also
 *   in tests/test_gpu_rccl_branch.py).  brick2_split is gone (the launcher decides from the number of bricks).
 *   debug_ghost_cap, debug_early_reuse: planted faults for tests (test_an_outgrown_capacity_is_redone_not_fatal; tests/test_gpu_rccl_branch.py).
 *   check_launches 0  debugging: every stage of a rebuild (migration, reorder, borders, list builder) is synchronised and asked for
 *                    HIP errors, so that a fault names the stage instead of surfacing at the end of meso_run
 *   profile       0  HIP-event timers per phase (meso_timer_get); pair_debug: timing ablations (bench only) */
int meso_set_option(meso_ctx *ctx, const char *key, double value);

/* ---- domain + decomposition: Domain box, MesoComm procgrid (comm_meso.cu:41-186, src/comm.cpp Comm::setup) */
int meso_set_box(meso_ctx *ctx, const double boxlo[3], const double boxhi[3], const int periodicity[3]);
/* one rank per GPU on a brick procgrid; must precede meso_atoms_upload.  uid = 128-byte ncclUniqueId from rank 0
 * (RCCL), or an 8-byte group id shared by the ranks of one process (LOCAL: several contexts, one host thread each) */
int meso_comm_init(meso_ctx *ctx, int nranks, int rank, const int procgrid[3], int transport, const void *uid,
                   size_t uid_bytes);
int meso_comm_get_unique_id(void *uid, size_t uid_bytes);
/* brick processor grid of minimal surface for the box (Comm::set_procs, src/comm.cpp); host only, no GPU needed */
int meso_decomp_procgrid(int nranks, const double prd[3], int procgrid[3]);
/* the decomposition rank `rank` (x fastest in the grid) works with - sub-box (Domain::set_local_box, src/domain.cpp), slabs within
 * cutghost of the faces that have a neighbour, and per direction dir = (sx+1) + 3(sy+1) + 9(sz+1): owner of the neighbouring sub-box,
 * whether anything is sent that way, the periodic shift applied on the way and the centre of the neighbour's sub-box
 * (Comm::setup, src/comm.cpp; MesoComm::borders comm_meso.cu:41-186 walks the same table).  The engine uses this very function;
 * host only, no GPU needed */
int meso_decomp_plan(const double boxlo[3], const double boxhi[3], const int periodic[3], const int procgrid[3], int rank,
                     double cutghost, double sublo[3], double subhi[3], double slab_lo[3], double slab_hi[3], int peer27[27],
                     int active27[27], double shift27[81], double center27[81]);
/* ranks the transport really connects: ncclCommCount for RCCL (bench.py's n_ranks_seen), the configured count otherwise */
int meso_comm_count(meso_ctx *ctx, int *nranks_seen);
/* host-staged transport (tests: several ranks sharing one GPU): exchange(user, npeer, peer[], sendbuf[],
 * sendbytes[], recvbuf[], recvbytes[]) must deliver every buffer; all pointers are host memory */
typedef int (*meso_host_exchange_fn)(void *user, int npeer, const int *peer, const void *const *sendbuf,
                                     const size_t *sendbytes, void *const *recvbuf, const size_t *recvbytes);
int meso_comm_set_host_exchange(meso_ctx *ctx, meso_host_exchange_fn fn, void *user);

/* ---- atoms: AtomStyle(dpd/atomic/meso,AtomVecDPDAtomic) atom_vec_dpd_atomic_meso.h:3; upload replaces
 *      MesoAtomVec::transfer CUDACOPY_C2G (atom_vec_meso.cu:220-323); x,v are LAMMPS AoS double[n][3] */
int meso_set_mass(meso_ctx *ctx, int ntypes, const double *mass_per_type /* [ntypes+1], index 0 unused */);
int meso_atoms_upload(meso_ctx *ctx, int nlocal, const double *x, const double *v, const int *tag,
                      const int *type, const int *mask /* nullable: all 1 */, const int *image /* nullable */);
int meso_atoms_count(meso_ctx *ctx, int *nlocal, int *nghost, int *n_bulk);
/* transfer_pre_output (atom_meso.cu:258-266): device order; any pointer may be NULL; arrays sized nlocal */
int meso_atoms_download(meso_ctx *ctx, double *x, double *v, double *f, int *tag, int *type, int *image);

/* ---- neighbor / neigh_modify commands (src/neighbor.cpp Neighbor::set / modify_params) */
int meso_neighbor(meso_ctx *ctx, double skin, int every, int delay, int check);

/* ---- pair_style dpd/meso rc seed ; pair_coeff i j a0 gamma sigma s [rc]  (pair_dpd_meso.cu:272-327) */
int meso_pair_dpd_settings(meso_ctx *ctx, int style, double cut_global, int seed);
/* MESO_PAIR_DPD_POLYFORCE: PairStyle(dpd/polyforce/meso) pair_dpd_polyforce_meso.h:3 - the fp32 kernel with the conservative
 * force polyval(1 - r/rc) (Horner, c_order first; kernel :159-162, energy :176); pair_coeff i j gamma sigma order c_order..c_0
 * (:290-335), cutoff = the style's global cutoff, order < 32 */
int meso_pair_dpd_polyforce_coeff(meso_ctx *ctx, int itype, int jtype, double gamma, double sigma, int order, const double *c);
/* MESO_PAIR_DPD_TABLEFORCE: PairStyle(dpd/tableforce/meso) pair_dpd_tableforce_meso.h:3 - the fp32 kernel with the conservative
 * force read from a table of table_length points uniform in r/rc over [0,1] (linear filter of the texture fetch :181, :291,
 * weights to 8 fractional bits) and uniform TEA noise (:171); pair_coeff i j gamma sigma <table> (:306-356; the glue reads a
 * file argument on the host).  No pair energy is booked (as in the reference kernel). */
int meso_pair_dpd_tableforce_coeff(meso_ctx *ctx, int itype, int jtype, double gamma, double sigma, int table_length,
                                   const double *table);
int meso_pair_dpd_coeff(meso_ctx *ctx, int itype, int jtype, double a0, double gamma, double sigma, double expw,
                        double cut /* <=0: cut_global */);

/* ---- bonded topology (configs[4]): AtomStyle(dpd/bond/meso) atom_vec_dpd_bond_meso.cu:20-45, BondStyle(harmonic/meso)
 *      bond_harmonic_meso.cu:46-117, exclusions neigh_build_meso.cu:497-569.  Call after meso_atoms_upload, with the
 *      whole Bonds section on every rank; special_bonds first (weights 0 or 1, default 0 0 0 like src/force.cpp:47-48) */
int meso_special_bonds(meso_ctx *ctx, double w12, double w13, double w14);
int meso_bonds_upload(meso_ctx *ctx, int nbonds, const int *tag_i, const int *tag_j, const int *bond_type);
int meso_bond_style_harmonic(meso_ctx *ctx, int nbondtypes);
int meso_bond_coeff(meso_ctx *ctx, int type, double k, double r0);
/* BondStyle(fene/meso) bond_fene_meso.h:3, coefficients K R0 epsilon sigma (bond_fene_meso.cu:36-50; FENE + WCA, the
 * log argument clamped at 0.1 as in gpu_bond_fene :103-109) */
int meso_bond_style_fene(meso_ctx *ctx, int nbondtypes);
int meso_bond_coeff_fene(meso_ctx *ctx, int type, double k, double r0, double epsilon, double sigma);
int meso_bond_compute(meso_ctx *ctx, int eflag);          /* Bond::compute, adds to f */
int meso_compute_ebond(meso_ctx *ctx, double *e_total);
/* AtomStyle(dpd/angle/meso) atom_vec_dpd_angle_meso.h:3, AngleStyle(harmonic/meso) angle_harmonic_meso.h:3 (kernel
 * angle_harmonic_meso.cu:46-172, tag mapping neighbor_meso.cu:161-182).  The Angles section (tag1, apex tag2, tag3, type)
 * follows the Bonds section; angle_coeff type K theta0[degrees] as in src/MOLECULE/angle_harmonic.cpp:157-181.
 * meso_run / meso_setup evaluate bonds and angles themselves; meso_angle_compute is Angle::compute for a host-driven step. */
int meso_angles_upload(meso_ctx *ctx, int nangles, const int *tag1, const int *tag2, const int *tag3, const int *angle_type);
int meso_angle_style_harmonic(meso_ctx *ctx, int nangletypes);
int meso_angle_coeff(meso_ctx *ctx, int type, double k, double theta0_degrees);
int meso_angle_compute(meso_ctx *ctx, int eflag);
int meso_compute_eangle(meso_ctx *ctx, double *e_total);

/* ---- timestep, fix nve/meso group */
int meso_timestep(meso_ctx *ctx, double dt);

/* ---- integrator: IntegrateStyle(mvv/meso,ModifiedVerlet) mvv_meso.h:3-4 */
int meso_setup(meso_ctx *ctx);                 /* ModifiedVerlet::setup  mvv_meso.cu:139-219 */
int meso_run(meso_ctx *ctx, int nsteps);       /* ModifiedVerlet::run    mvv_meso.cu:243-425 */
/* the individual virtuals the LAMMPS host calls, for glue code that keeps its own run loop */
int meso_nve_initial(meso_ctx *ctx);           /* FixNVEMeso::initial_integrate fix_nve_meso.cu:97-155 */
int meso_nve_final(meso_ctx *ctx);             /* FixNVEMeso::final_integrate   fix_nve_meso.cu:180-199 */
int meso_neighbor_decide(meso_ctx *ctx, int *rebuild); /* Neighbor::decide src/neighbor.cpp:1216-1231 */
int meso_reneighbor(meso_ctx *ctx);            /* pbc+exchange+sort_local+borders+build mvv_meso.cu:270-335 */
int meso_halo_forward(meso_ctx *ctx);          /* Comm::forward_comm + transfer_pre/post_comm mvv_meso.cu:338-358 */
int meso_force_clear(meso_ctx *ctx, int range);/* MesoAtomVec::force_clear atom_vec_meso.cu:325-336 */
int meso_pair_compute(meso_ctx *ctx, int range, int eflag, int vflag); /* Pair::compute/compute_bulk/compute_border
                                                                          pair_dpd_meso.cu:241-266 */
int meso_step_advance(meso_ctx *ctx, int64_t ntimestep); /* update->ntimestep for glue-driven loops */

/* ---- computes: ComputeStyle(temp/meso,MesoComputeTemp) compute_temp_meso.cu:77-101; pe/pressure
 *      compute_pe_meso.cu:66-125 */
int meso_compute_temp(meso_ctx *ctx, double *temperature);
int meso_compute_pe(meso_ctx *ctx, double *pe_total);
int meso_compute_pressure(meso_ctx *ctx, double *pressure);
/* a thermo step that was not announced to the force call (Integrate::ev_set src/integrate.cpp:117-150 sets eflag/vflag BEFORE
 * the step's force; ModifiedVerlet::run mvv_meso.cu:411-415 prints after it): per-atom energy and virial are tallied at the
 * current configuration with the forces the run continues with left untouched; no-op when this step's force call tallied */
int meso_tally_ev(meso_ctx *ctx);

/* ---- introspection used by the parity tests and bench.py */
int meso_neigh_info(meso_ctx *ctx, int *n_col, int *max_count, double *avg_count, int64_t *nbuild);
/* row-major copy of the neighbour table: table[i*stride + p], rows of atoms in device order */
int meso_neigh_download(meso_ctx *ctx, int *count, int *table, int stride);
int meso_merged_download(meso_ctx *ctx, float *coord4, float *veloc4, int nall);
/* partitioned rows (round 5; the reference keeps its rows in two sections as well - "core from the front, skin from the back",
 * neigh_build_meso.cu:91-115 - split by distance; here by who evaluates a pair): *parted = 1 when the table in use holds, per atom,
 * a front section (the pairs this atom evaluates: partners outside its aligned group of `group` atoms one-sided, partners inside
 * it for both atoms) and a back section (in-group partners that evaluate the pair themselves), each padded with the atom itself
 * to a multiple of 8 entries.  meso_neigh_download returns the neighbours alone, front section first.
 * nfront, nback (nullable): [nlocal] section lengths; front, back (nullable): the sections as stored, padding included,
 * front[i * stride + p] / back[i * stride + p]. */
int meso_neigh_parts(meso_ctx *ctx, int *parted, int *group, int *nfront, int *nback, int *front, int *back, int stride);
/* per-phase device time (ms, HIP events on the engine's stream) accumulated since the last reset;
 * names: "pair","neigh","nve","merge","halo","reorder","bin","total_steps" */
int meso_timer_reset(meso_ctx *ctx);
int meso_timer_get(meso_ctx *ctx, const char *name, double *ms, int64_t *calls);
int64_t meso_ntimestep(meso_ctx *ctx);
/* measured HBM peak for the roofline (nominal and measured): float4 copy of nbytes (read + write counted),
 * best of reps launches timed with HIP events on the engine's stream; result in GB/s */
int meso_membw_probe(meso_ctx *ctx, size_t nbytes, int reps, double *copy_gbs);
/* measurement only - floors of the fp32 force kernel's mandatory work on the table in use (gpu_dpd_fast<0>, pair_dpd_fast_meso.cu:124-162:
 * per row entry a coordinate gather and a cutoff test, per pair inside the cutoff a velocity gather and the TEA / Gaussian / force
 * evaluation), each timed alone in an idealised kernel: mode 1 the arithmetic, 2 the loads, 3 both, independent (pair_floor.hip).
 * us = mean launch time over reps; counts[0] = row entries walked, counts[1] = pairs evaluated.  Overwrites the force arrays. */
int meso_pair_floor(meso_ctx *ctx, int mode, int reps, double *us, long long *counts);
/* name of the force-kernel instantiation the last launch ran ("k_pair_dpd_ring<true, 0, true, true, 1, true>", as rocprofv3
 * prints it; empty before the first launch): measurement harnesses key profile-derived numbers on it (the reference keeps
 * one static GridConfig per kernel instead, pair_dpd_meso.cu:216,228) */
int meso_pair_kernel_name(meso_ctx *ctx, char *buf, int nbuf);
/* host-side account of the exchanges of the host / in-process transports since meso_timer_reset (option profile 1): one line
 * per kind of exchange, "what|calls|ms until the device had the messages ready|ms on the wire incl. waiting for the slowest
 * peer|ms until the received bytes were back on the device|bytes sent" (the reference stamps Timer::COMM around the same calls,
 * mvv_meso.cu:289,320).  RCCL exchanges are stream-ordered and appear in a kernel trace instead */
int meso_xchg_stats(meso_ctx *ctx, char *buf, int nbuf);

/* ---- restart files and profiler window.  Inside LAMMPS the restart file is LAMMPS' own (the glue's
 *      Pair::write_restart keeps MesoPairDPD::write_restart's layout, pair_dpd_meso.cu:363-447).  The stand-alone driver
 *      writes one file per rank (path, or path.<rank> with several ranks) holding settings, coefficients and the per-atom
 *      arrays as they live on the device, forces included; meso_read_restart + meso_setup continue a run bit for bit when the
 *      file was written on a neighbour-rebuild step.
 *      meso_profile_window: MesoDevice::configure_profiler (engine_meso.cu:155-177); mode 0 off, 1 all, 2 core (the middle
 *      half of a run), 3 loop (the whole run), 4 interval [start, end) in absolute timesteps - collection of an attached
 *      rocprofv3 is paused outside the window (roctxProfilerPause/Resume, looked up at run time). */
enum { MESO_PROFILE_OFF = 0, MESO_PROFILE_ALL = 1, MESO_PROFILE_CORE = 2, MESO_PROFILE_LOOP = 3, MESO_PROFILE_INTERVAL = 4 };
int meso_write_restart(meso_ctx *ctx, const char *path);
int meso_read_restart(meso_ctx *ctx, const char *path);
int meso_profile_window(meso_ctx *ctx, int mode, int64_t start_step, int64_t end_step);

/* ---- known-answer kernels (math_meso.h:444-484) on caller-provided host arrays */
int meso_test_tea(meso_ctx *ctx, int n, int rounds, const uint32_t *u, const uint32_t *v, uint32_t *out0,
                  uint32_t *out1);
int meso_test_gaussian(meso_ctx *ctx, int n, const uint32_t *u, const uint32_t *v, double *out_dp, float *out_sp);
int meso_test_logistic(meso_ctx *ctx, int n, const uint32_t *u, const uint32_t *v, float *out); /* mean0var1<8> pair_dpd_minimal_meso.cu:82-89 */
uint32_t meso_seed_now(int seed, int64_t ntimestep); /* MesoPairDPD::seed_now pair_dpd_meso.cu:268-270 */

/* ---- mini driver: runs the input-script subset of example/simple/{sp,dp}.run unchanged */
int meso_script_run(meso_ctx *ctx, const char *path, const char *var_name, const char *var_value, char *log,
                    size_t log_bytes);

#ifdef __cplusplus
}
#endif
#endif
