// Host-side engine of the MI355X DPD hot path: device-resident particle state, the per-step schedule
// of run_style mvv/meso, neighbour rebuilds, halo exchange.  Mirrors the roles of the reference's
// MesoDevice / MesoAtom(+Vec) / MesoNeighbor / MesoComm / ModifiedVerlet behind one
// context object; see include/meso_hip.h for the C ABI that exposes it.
#pragma once
#include "kernels.h"
#include "sort.h"
#include <hip/hip_runtime.h>
#include <map>
#include <string>
#include <vector>

struct ncclComm;

namespace meso {

struct FusedArgs;
struct DirTab;
struct LocalGroup;

struct PhaseTimer {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    double ms = 0.0;
    long calls = 0;
};

typedef int (*host_exchange_fn)(void *user, int npeer, const int *peer, const void *const *sendbuf,
                                const size_t *sendbytes, void *const *recvbuf, const size_t *recvbytes);

class Engine {
public:
    explicit Engine(int device);
    ~Engine();

    // configuration
    int set_box(const double *lo, const double *hi, const int *per);
    int comm_init(int nranks, int rank, const int *procgrid, int transport, const void *uid, size_t uid_bytes);
    int set_mass(int ntypes, const double *mass);
    int atoms_upload(int n, const double *x, const double *v, const int *tag, const int *type, const int *mask,
                     const int *image);
    int atoms_download(double *x, double *v, double *f, int *tag, int *type, int *image);
    int neighbor(double skin, int every, int delay, int check);
    int bonds_upload(int nbonds, const int *tag_i, const int *tag_j, const int *btype);
    int special_bonds(double w12, double w13, double w14);
    int bond_style(int nbondtypes, int kind = 0);      // kind 0: harmonic/meso, 1: fene/meso
    int bond_coeff(int type, double k, double r0, double eps = 0.0, double sigma = 0.0);
    int bond_compute(int eflag, int store = 0);
    int compute_ebond(double *e);
    // restart.hip: per-rank restart files of the stand-alone driver, profiler window (-profile all|core|loop|interval)
    int write_restart(const std::string &path);
    int read_restart(const std::string &path);
    // what the script driver needs to know after read_restart
    int restart_ntypes() const { return ntypes; }
    const std::vector<double> &restart_masses() const { return mass_type; }
    void restart_box(double *lo, double *hi, int *per) const
    {
        for (int d = 0; d < 3; d++) { lo[d] = boxlo[d]; hi[d] = boxhi[d]; per[d] = periodic[d]; }
    }
    int profile_window(int mode, int64_t start, int64_t end);
    void profile_tick(int it, int nsteps);
    int prof_mode = 0, prof_windows = 0;
    int64_t prof_start = 0, prof_end = 0;
    int (*prof_pause)(uint64_t) = nullptr;
    int (*prof_resume)(uint64_t) = nullptr;
    bool wrap_in_reorder = false;     // this rebuild's periodic wrap is done by the reorder's key kernel
    bool merged_in_reorder = false;   // coord4/veloc4 of the locals were written by the reorder gather of this rebuild
    bool restart_forces = false;  // set by read_restart, consumed by setup
    bool upload_all = false;      // read_restart: atoms_upload keeps every atom it is given
    // Angles section + angle_style harmonic/meso (atom_style dpd/angle/meso, angle_harmonic_meso.cu)
    int angles_upload(int na, const int *t1, const int *t2, const int *t3, const int *type);
    int angle_style(int nangletypes);
    int angle_coeff(int type, double k, double theta0_deg);
    int angle_compute(int eflag);
    int compute_eangle(double *e);
    int pair_settings(int style, double cut, int seed);
    int pair_coeff_poly(int i, int j, double gamma, double sigma, int order, const double *c);
    int pair_coeff_table(int i, int j, double gamma, double sigma, int len, const double *t);
    int pair_coeff(int i, int j, double a0, double gamma, double sigma, double expw, double cut);
    int set_option(const std::string &key, double val);

    // schedule
    int setup();
    int run(int nsteps);
    int nve_initial();
    int nve_final();
    int decide(int *rebuild);
    int reneighbor();
    int halo_forward();
    int force_clear(int range);
    int pair_compute(int range, int eflag, int vflag);
    int tally_ev();
    void launch_pair(PairArgs &p, int ev);
    bool launch_refused = false;    // the ring kernel's launcher declined a launch (reported by run() / pair_compute as an error)
    bool ring_selected() const;
    char pair_variant[128] = "";    // instantiation of the force kernel THIS engine launched last (meso_pair_kernel_name)

    // computes
    int compute_temp(double *t);
    int compute_pe(double *pe);
    int compute_pressure(double *p);

    // introspection
    int neigh_info(int *n_col, int *max_count, double *avg, int64_t *nbuild);
    int neigh_download(int *count, int *table, int stride);
    int neigh_parts(int *parted, int *group, int *nfront, int *nback, int *front, int *back, int stride);
    int merged_download(float *c4, float *v4, int nall);
    int timer_reset();
    int timer_get(const std::string &name, double *ms, int64_t *calls);
    int test_tea(int n, int rounds, const uint32_t *u, const uint32_t *v, uint32_t *o0, uint32_t *o1);
    int test_gaussian(int n, const uint32_t *u, const uint32_t *v, double *odp, float *osp);
    int test_logistic(int n, const uint32_t *u, const uint32_t *v, float *out);
    int sync();
    int resolve_counts();      // counts of the last rebuild that are still on their way to the host (async_counts)
    int comm_count(int *n);
    // host-side account of the exchanges of the host / in-process transports while option profile is on (comm.hip xchg)
    struct XchgStat { long calls = 0; double ms_device = 0, ms_wire = 0, ms_back = 0, bytes = 0; };
    std::map<std::string, XchgStat> xchg_stats;
    std::string xchg_report();
    // RCCL exchanges are stream-ordered: with option profile their groups are bracketed by HIP events on the exchange stream and
    // booked (device time between the events) when the report is asked for
    struct XchgEvent { hipEvent_t a, b; std::string what; double bytes; };
    std::vector<XchgEvent> xchg_events;
    void xchg_events_flush();
    bool rccl_first_done = false;   // the first RCCL group of this communicator has completed (bounded wait, comm.hip xchg)
    int membw_probe(size_t nbytes, int reps, double *gbs);
    int pair_floor(int mode, int reps, double *us, long *counts);      // pair_floor.hip: measured floors of the force kernel's mandatory work

    std::string err;
    int64_t ntimestep = 0;
    int nlocal = 0, nghost = 0, n_bulk = 0;
    double dt = 0.005;
    host_exchange_fn host_exchange = nullptr;
    void *host_exchange_user = nullptr;

private:
    int fail(int code, const std::string &msg);
    int check(hipError_t e, const char *what);
    int init_params();
    int ensure_capacity(int need_atoms);
    int alloc_atoms(int cap);
    void free_all();
    void range(int r, int &beg, int &end) const;
    int merge_locals(uint32_t seed);
    int halo_borders();
    int halo_forward_seed(uint32_t seed, bool async = false);
    int build_cells_and_table();
    int reorder_locals();
    int migrate();
    int check_overflow();
    // a rebuild whose capacities were outgrown is redone inside run() (PairArgs::poison keeps the state; engine.hip prepare_redo)
    static constexpr int MESO_REDO = -7001, MESO_DEEPER = -7002;
    bool in_reneighbor = false;
    int reneighbor_once();
    int prepare_redo(int code);
    bool redo_armed = false, ck_swapped = false;
    int ck_nlocal = 0;
    long nredo = 0;              // rebuilds redone so far (meso_neigh_info-style introspection for the tests: option query below)
    int debug_early_reuse = 0;   // option (tests): the refresh's send staging is scribbled from an unordered stream right behind the exchange
    hipStream_t debug_stream = nullptr;
    hipEvent_t debug_event = nullptr;
    int debug_ghost_cap = 0;     // option (tests): the NEXT asynchronous rebuild reserves this many ghosts only
    double reduce_global_sum(double v);
    // multi-rank (comm.hip)
    int xchg(int np, const int *peer, void *const *sbuf, const size_t *sbytes, void *const *rbuf, const size_t *rbytes, void *const *rbuf2 = nullptr);
    int merge_new_ghosts(uint32_t sd);
    // several ranks, borders without a host round trip (comm.hip): fixed-capacity messages, counts in band
    std::vector<int> mr_cap_s, mr_cap_r;          // ghosts per peer message, from the counts of the previous rebuild (same on both sides)
    bool mr_caps_ready = false, mr_pending = false;
    // option: capacity = count * (1 + margin) + 256.  A border message is sent at its capacity, so head-room costs wire bytes
    // (64 B per ghost, once per rebuild); what it must cover: between two rebuilds an atom moves at most D (the neighbour list is
    // only valid while 2 D <= skin), so the ghosts of a slab of width r_c + skin can grow by at most the atoms of a layer D thick
    // next to it: D / (r_c + skin) <= 11.5 % at equal density, times the density contrast across the slab's edge.  0.5 covers a
    // contrast of 4 (phase-separating decks); a message that still outgrows it ends the job with an error on every rank
    // (the launcher tears the job down on the first rank's status), never with a truncated ghost list.
    double mr_cap_margin = 0.5;
    int *d_mr = nullptr;                          // device-side offsets and counts of the exchange in flight (128 ints)
    // migration with the counts in the messages (comm.hip)
    std::vector<int> mig_cap_s, mig_cap_r;
    bool mig_caps_ready = false;
    int mig_cap_floor = 64;                       // option: capacity of a migration message = 2 * previous count + floor
    long mig_resends = 0;                         // messages that had to be sent again (statistics)
    void *stage2_send = nullptr, *stage2_recv = nullptr;      // exact resend of a migration message that outgrew its capacity
    size_t stage2_send_bytes = 0, stage2_recv_bytes = 0;
    void mig_update_caps(const std::vector<int> &send_n, const std::vector<int> &recv_n);
    int migrate_inband();
    bool mr_async_ok() const;
    void mr_update_caps();
    int halo_borders_multi_async();
    int mr_resolve();
    const int *pending_nghost_dev() const { return !counts_pending ? nullptr : (mr_pending ? d_mr + 64 : d_dir_start + 27); }
    int exchange_counts(int skip_stay, int *h_ds, std::vector<int> &send_n, std::vector<int> &recv_n, std::vector<int> &recv_dir);
    void build_peer_tables();
    int halo_borders_multi();
    int halo_forward_multi_begin(uint32_t seed, bool async);
    int halo_wait();
    int ensure_stage(size_t sbytes, size_t rbytes);
    bool owns(const double *x) const;
    void comm_free();
    DirTab &fwd_tab_host();
    void free_fwd_tab();
    void tbegin(const char *name);
    void tend(const char *name);
    void tflush();

    int device;
    hipStream_t stream = nullptr, side = nullptr;
    bool profiling = false;
    std::map<std::string, PhaseTimer> timers;
    std::vector<hipEvent_t> event_pool;

    // box / decomposition
    double boxlo[3], boxhi[3], prd[3];
    int periodic[3];
    bool have_box = false;
    int nranks = 1, rank = 0, procgrid[3] = {1, 1, 1}, myloc[3] = {0, 0, 0};
    int transport = 0;
    ncclComm *nccl = nullptr;
    LocalGroup *local = nullptr;
    std::vector<int> peers, peer_send_n, peer_recv_n, peer_send_base, peer_recv_base;
    int peer_index[27];
    void *stage_send = nullptr, *stage_recv = nullptr;
    size_t stage_send_bytes = 0, stage_recv_bytes = 0;
    int *sendlist_aux = nullptr;
    const char *xchg_what = "ghost refresh";     // names the exchange in transport error messages
    DirTab *fwd_tab = nullptr;
    hipStream_t xs = nullptr;       // stream used by xchg (nullptr: the main stream)
    hipEvent_t ev_pack = nullptr, ev_halo = nullptr;
    int overlap = 1;                // bulk force kernel overlaps the ghost refresh (nranks > 1)
    double sublo[3], subhi[3];
    double slab_lo[3], slab_hi[3];
    double shift27[81], center27[81];
    int peer27[27];
    bool send_active[27];

    // settings
    double skin = 0.3;
    int every = 1, delay = 10, dist_check = 1;
    int groupbit = 1;
    int pair_style = 0, seed = 0, ntypes = 0;
    bool pair_ftab = false;  // pair_style dpd/tableforce/meso: fp32 arithmetic, tabulated conservative force, uniform TEA noise
    int ftab_len = 0;
    std::vector<float> ftab; // [ntypes^2][ftab_len]
    float *d_ftab = nullptr;
    bool pair_poly = false;  // pair_style dpd/polyforce/meso: fp32 arithmetic, polynomial conservative force
    std::vector<float> poly; // [ntypes^2][MESO_POLY_PITCH]
    float *d_poly = nullptr;
    int tile_plan = 0;       // option: 1 = separate k_brick_plan launch before the tile builder (former path), 0 = plan inside it
    int reorder_sort = 0;    // option: 1 = reorder the locals with the radix/merge sort (the former path), 0 = by counting
    int *rcount = nullptr;   // [2M+1] atoms per extended code
    int reorder_cap_user = 0;
    int reorder_cap = 2048;  // LDS stage of k_reorder_order (pairs per 128 codes), from the density
    int ghost_sort = 0;      // option: 1 = bin the ghosts with the radix/merge sort (the former path), 0 = by counting
    int *gcount = nullptr;   // [M+1] ghosts per Morton code
    int pair_npart = 0;      // option pair_npart: lanes per atom in the ring kernel (0 = by launch size)
    int pair_rng = 0;       // 1: pair_style dpd/mini/meso (fp32 arithmetic of dpd/fast/meso, logistic-map noise, one coefficient set)
    double cut_global = 0.0, cutmax = 0.0, cutghost = 0.0;
    bool have_pair = false, have_coeff = false, params_ready = false, is_setup = false;
    std::vector<double> coeff;      // ntypes*ntypes*7
    std::vector<int> coeff_set;
    std::vector<double> mass_type;  // ntypes+1
    int neigh_kernel = 1;           // 0 simple, 1 wave/LDS
    int pair_kernel = 2;            // 0 lane-per-atom, 1 tile/brick, 2 auto (fp32 on cell rows: ring, fp64: MLP + compaction),
                                    // 3 MLP + ballot compaction, 4 MLP only
    int pair_debug = 0;             // timing ablations (bench only)
    int fuse_bonds = 1;             // option: bond forces inside the force kernel's step-boundary epilogue (systems without angles)
    int pair_share = 1;             // fp32 ring kernel: pairs inside one aligned 256-atom group are evaluated once
    int fuse_pair = 1;              // step boundary in the epilogue of the fp32 ring kernel (no separate NVE pass, forces not stored)
    int fuse_step = 1;              // final(s)+initial(s+1)(+merge) in one kernel between steps of one run()
    int fuse_clear = 1;             // pair kernel writes f instead of clear + accumulate
    long natoms_total = 0;

    // atoms (device)
    int nmax = 0;
    AtomSoA cur{}, alt{};
    float4 *coord4 = nullptr, *veloc4 = nullptr;
    float4 *coord4_next = nullptr, *veloc4_next = nullptr;   // written by the force kernel's step-boundary epilogue, then swapped in
    double *virial[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    double *e_pair = nullptr;
    double *xhold = nullptr;
    double *d_mass_type = nullptr, *d_coeff64 = nullptr;
    double *d_dtfm_type = nullptr;  // 0.5 dt / mass per type (launch_dtfm_table) for the step-boundary epilogue; dtfm_for: the 0.5 dt it holds
    double dtfm_for = -1.0;
    float *d_coeff32 = nullptr;

    // neighbour
    BinGeom geom{};
    int n_col = 0;
    int *pair_count = nullptr, *pair_table = nullptr;
    size_t table_tiles = 0;
    uint32_t *bin_key = nullptr, *bin_key_alt = nullptr;      // ghost binning scratch
    int *bin_val = nullptr, *bin_val_alt = nullptr;
    int64_t nbuild = 0;
    int ago = 0;

    // brick layout
    static constexpr int layout = 2;   // cell order = storage order (the bin-sorted and brick layouts of round 1 were retired)
    int *estart = nullptr, *gstart = nullptr, *gslot = nullptr;
    int4 *binrange = nullptr;
    int *brick_hoff = nullptr, *brick_hdr = nullptr;
    uint32_t *brick_hmap = nullptr, *brick_own = nullptr;
    size_t brick_cap = 0;
    int brick_maxh_alloc = 0;
    bool bulk_pending = false;      // n_bulk of the last reorder is still on its way to the host
    // rebuilds without a host round trip (one rank, cell-ordered layout): launch sizes come from the previous rebuild's counts
    // plus head-room, kernels mask with the counts on the device, the host reads them when it next needs them
    int async_counts = 1;           // option
    double async_grid_scale = 1.125;   // grids of the ghost kernels: previous ghost count x this + 1024 (they loop: any count is covered)
    bool counts_pending = false;
    hipEvent_t ev_counts = nullptr;
    int report_seq = 0;             // one rank, fused rebuild: the report carries a sequence number the host polls (no event between the kernels)
    bool counts_by_seq = false;
    int report_poll = 1;            // option: 0 = an event behind the rebuild's last launch, as before
    int nghost_prev = -1, n_bulk_prev = -1;
    bool async_ok() const;
    // ... and with the two halves of the rebuild on two streams (option overlap_rebuild): the reorder of the locals on the
    // main stream, border lists + ghost creation + ghost binning on the side stream (north_star: reorder on a side stream
    // overlapped with halo pack/unpack; the reference overlaps its sort-phase transfers, mvv_meso.cu:296-316)
    // one rank, small boxes: the step-boundary epilogue also refreshes the ghosts (no k_pack_forward launch between rebuilds)
    int ghost_epilogue = -1;        // option: -1 on (up to 524288 local atoms without xcd_balance: with every border atom on the last XCD the image
                                    // writes lengthened the 64^3 launch by 10 us; dealt out over the XCDs they cost 2 us and save the refresh kernel), 0 off, 1 on
    int *img_cnt = nullptr, *img = nullptr;
    double *d_shift27 = nullptr;
    bool images_ready = false;      // this rebuild recorded the images
    bool build_images_now = false;
    // (an atom has at most 7 periodic images - the table holds 8 - only while every periodic edge exceeds twice the ghost cutoff)
    bool img_ok = false;
    bool images_on() const { return nranks == 1 && img_ok && (ghost_epilogue == 1 || (ghost_epilogue < 0 && (nlocal <= 524288 || xcd_balance))); }
    // one rank: the rebuild in three launches (rebuild.hip) - count, place + gather + ghost emission, ghosts
    int *brick_order2 = nullptr;    // launch order of the 2-bricks (those that own real cells, fullest first)
    int brick2 = 1;                 // option: 2x2x2 bricks in the list builder (0: the 4x4x4 bricks of rounds 1-2)
    int tile_persist = 0;           // option: the 2-brick list builder as persistent workgroups that draw bricks from counters (launches of more than one
                                    // round; 64^3: 219 against 206 us per build with one workgroup per brick - profiles/r06_notes.md section 1)
    int *tile_queue = nullptr;      // [2] next brick, workgroups done (put back to zero by the last workgroup of every launch)
    int brick2_limit = 1 << 30;     // ... while the bin grid spans at most this many 4-bricks (measured faster at every size:
                                    // 32^3 77 -> 52 us, 48^3 205 -> 137, 64^3 303 -> 265, 128^3 2257 -> 1784 per build)
    bool brick2_off = false;        // a 2-brick neighbourhood outgrew the largest stage that leaves five workgroups per CU: 4-bricks from then on
    int refresh_epilogue = 1;       // option: several ranks - the per-step refresh messages are written by the force kernel's step boundary (image
                                    // tables from the rebuild's border kernel) instead of k_pack_forward_multi
    bool mr_images_ready = false;   // ... this rebuild recorded the tables
    bool fwd_packed = false;        // ... the last force launch wrote the next refresh
    void *mr_img_stage = nullptr;   // (the staging the tables point into: a regrown staging voids them)
    int *d_vofs = nullptr;
    double *d_center27 = nullptr;
    bool mr_img_wanted() const;
    unsigned img_alloc_gen = 0, img_zero_gen = 0;
    int refresh_direct = 1;         // option: the per-step ghost refresh is received straight into the merged arrays when the ghosts are in message order
    int border_fused = 1;           // option: border lists + headers + records in one launch behind count + scan (0: fill, header, pack)
    int mig_slim = 1;               // option: leavers' lists by atomics + ranking (2 launches) instead of the counting chain (8)
    bool mig_lists_built = false;   // this rebuild's direction-major list of all atoms exists (sendlist, d_dir_start)
    int *mig_cnt = nullptr, *mig_lst = nullptr;
    int mig_lst_n = 0;
    int build_mig_lists();
    bool mig_slim_now() const;
    bool novf_pending = false;      // the fused reorder left its overflow count for the border count kernel to clear
    bool mig_holes = false;         // several ranks: the migration left its leavers in place (holes) and appended the arrivals behind the
    int mig_span = 0, mig_nold = 0; // old atoms: the reorder's count walks mig_span atoms and skips the leavers among the first mig_nold
    bool reorder_fuses(long n) const;
    int border_runs = 1;            // option: several ranks keep a rebuild's ghosts in message order and read the ghost cells as runs (comm.hip)
    bool mr_runs = false;           // ... this rebuild did
    int *mr_gcnt = nullptr;         // [M+1] ghosts per ghost cell of that form (gstart holds the starts)
    int mr_gcnt_n = 0;
    int brick2_floor = 0;           // the 2-brick's LDS stage (atoms) after it grew during the run
    // partitioned rows (RowPartArgs, kernels.h): the list builder decides the Newton pairing of in-group pairs once per rebuild
    int merge_ghosts = 1;           // option: one rank, split_gather: the ghost tiles run in the gather's launch (k_fr_gather_ghosts)
    int lean_boundary = 1;          // option: the force kernel's step-boundary epilogue takes type and mass from what it holds (NveArgs::mass_type)
    int fuse_count = 1;             // option: the rebuild's count kernel runs in the epilogue of the force launch in front of the rebuild
    bool count_in_epilogue = false; // ... and has done so for the rebuild that follows
    int prepare_count_in_epilogue(FrCountArgs &c, bool &ok);
    int xcd_balance = 1;            // option: the force launch deals bulk and border workgroups out over the XCDs separately (PairArgs::bulk_hint)
    int check_launches = 0;         // option (debugging): synchronise and ask for HIP errors after every stage of a rebuild
    int launch_check(const char *stage);
    int row_part = -1;              // option: 1 rows in two sections (front: what the atom evaluates, back: mirrored entries); 0 plain rows;
                                    // -1 by rebuild interval and style (build_cells_and_table)
    int *pair_nback = nullptr;      // [nmax] back entries per atom
    int *pair_back = nullptr;       // the back table: chunked-8 rows of nb_col entries
    int nb_col = 0;
    bool rows_part = false;         // the table in use is partitioned
    int part_group = 0;             // ... for this pairing group of the force kernel
    int split_gather = -1;          // option: the fused rebuild's placing kernel only orders and a streaming pass gathers (-1: boxes of >= 200 k atoms - 40^3 +1 %, 48^3 +1.8 %, 64^3 +1.6 %, 32^3 -0.7 %; 0; 1)
    int fused_rebuild = 1;          // option
    bool fused_active = false;      // this rebuild ran the fused path: ghosts sit in slot order, directions in senddir
    bool fused_dirty = false;       // a fused rebuild failed half-way: counters are cleared before the next one
    bool fused_ok() const;
    int rebuild_fused();
    void fused_locals_args(FusedArgs &a, bool ghost_stage = false);      // ghost_stage: the one-rank ghost stage follows - the count books the images
    int fused_alloc();
    unsigned long long *fr_bucket = nullptr, *fr_ovf = nullptr;
    int *fr_novf = nullptr, *fr_ttot[2] = {nullptr, nullptr}, *fr_stot[2] = {nullptr, nullptr};
    int *fr_gttot[2] = {nullptr, nullptr}, *fr_gstot[2] = {nullptr, nullptr};
    unsigned long long *fr_scratch = nullptr;
    int *fr_gorder = nullptr, *fr_gcnt = nullptr;     // ghost tiles that hold ghost cells; ghosts per cell
    int fr_ngorder = 0;
    bool fused_gcnt_valid = false;
    unsigned char *senddir = nullptr;
    int fr_cap = 0, fr_gcap = 0, fr_cap_want = 0, fr_cap_user = 0;
    size_t fr_M = 0;
    unsigned fr_epoch = 0;
    static constexpr int fr_ovf_cap = 65536;
    int overlap_rebuild = 0;        // measured slower at every size (profiles/r02_notes.md section 5): kept as a tested option
    bool ghosts_binned = false;     // this rebuild's ghosts were binned by rebuild_overlapped
    int rebuild_overlapped();
    void *scan_temp_side = nullptr;
    size_t scan_temp_side_bytes = 0;
    int *gtmp_placed = nullptr, *perm_inverse = nullptr;
    uint32_t *gtmp_code = nullptr;
    hipEvent_t ev_wrap = nullptr, ev_ghosts = nullptr;
    bool permute_forces = true;     // the reorder carries the forces along (not needed for the rebuilds inside run())
    bool tile_fits = true;          // the tile builder can stage a brick neighbourhood of this density in LDS
    double brick_margin = 1.0;      // multiplier on the expected halo population (inhomogeneous systems)
    bool regrow_only = false;       // init_params re-run for a larger brick stage on this rank alone (no collective)
    int brick_maxh_floor = 0;       // fullest brick neighbourhood seen + 8 %: the engine grows the LDS stage of the tile builder to it
    size_t estart_cap = 0;
    BrickArgs bargs{};
    int l1bits = 0;

    // reorder
    uint32_t *rkey = nullptr, *rkey_alt = nullptr;
    int *rval = nullptr, *rval_alt = nullptr;
    void *sort_temp = nullptr;
    size_t sort_temp_bytes = 0;

    // halo
    int *sendlist = nullptr;
    int send_cap = 0;
    int *chunk_count = nullptr, *chunk_offset = nullptr;
    int chunk_cap = 0;
    int *d_dir_start = nullptr;
    int h_dir_start[28];
    int nsend = 0;

    // bonded topology
    int bpa = 0, msp = 0, nbondtypes = 0, maxtag = 0, bond_kind = 0;
    int apa = 0, nangletypes = 0;
    bool have_angles = false;
    std::vector<double> angle_cf;            // [k(0..nat)][theta0 in radians]
    double *d_angle_cf = nullptr, *e_angle = nullptr;
    int *angle_idx = nullptr;                // [(i*apa+a)*3] mapped after every rebuild
    double special_w[3] = {0.0, 0.0, 0.0};
    std::vector<double> bond_kr0;            // [k(0..nbt)][r0(0..nbt)][epsilon][sigma]
    std::vector<int> h_tags;                 // tags of the atoms kept at upload (topology is attached by tag)
    double *d_bond_kr0 = nullptr, *e_bond = nullptr;
    int *bond_idx = nullptr, *tagmap = nullptr, *tagc = nullptr;
    // one bit per tag: referenced by a bond or an angle (only those enter the tag map of a rebuild); null: every tag does
    uint32_t *tagbits = nullptr;
    std::vector<uint32_t> h_tagbits;
    int upload_tagbits();
    bool have_bonds = false;
    int alloc_topology(AtomSoA &a, int cap, int keep);
    int rebuild_topology();
    int mig_stride() const { return 11 + (2 + 2 * bpa + msp + (apa > 0 ? 1 + 4 * apa : 0) + 1) / 2; }

    // scalars
    double *d_partial = nullptr, *d_scalar = nullptr;
    int *d_flags = nullptr;     // [0] overflow, [1] n_bulk
    int *h_flags = nullptr;     // pinned
    int *h_flags_dev = nullptr; // the same memory as the device sees it (kernels report counts straight to the host)
    double *h_scalar = nullptr; // pinned
    bool ev_valid = false;
};

int comm_unique_id(void *uid, size_t uid_bytes);
void decomp_procgrid(int nranks, const double *prd, int *pg);
int decomp_plan(const double *boxlo, const double *boxhi, const int *periodic, const int *procgrid, const int *myloc, double cutghost,
                double *sublo, double *subhi, double *slab_lo, double *slab_hi, int *peer27, int *active27, double *shift27,
                double *center27);
int script_run(Engine &E, const char *path, const char *var_name, const char *var_value, std::string &out);

} // namespace meso
