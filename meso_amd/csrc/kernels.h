// Launch wrappers for the HIP kernels of the DPD hot path (gfx950).  Raw device pointers and
// a stream only; no ownership.  Each wrapper names the reference kernel it replaces.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace meso {

struct BinGeom {
    double lo[3], hi[3];      // my sub-box
    double bininv[3];
    double binsize[3];
    int mbin[3];              // bins per dim incl. one ghost layer each side
    int nbin;
};

// SoA views (device pointers)
struct AtomSoA {
    double *x[3], *v[3], *f[3];
    int *tag, *type, *mask, *image;
    double *mass;
    // bonded topology, per atom (null / 0 for atom_style dpd/atomic/meso): bond_tag[i*bpa+b], special[i*msp+s]
    int *nbond, *bond_tag, *bond_type, *nspecial, *special;
    int bpa, msp;
    // angles the atom takes part in (all three atoms store an angle, newton off): angle_tag[(i*apa+a)*4 + {t1,t2,t3,type}]
    int *nangle, *angle_tag;
    int apa;
};

struct HaloShift { double prd[3]; };

// ---- atom kernels (atom_vec_meso.cu:142-167, fix_nve_meso.cu:62-95,157-178, compute_temp_meso.cu:58-75,
//      domain_meso.cu:30-145, memory_meso.h:17, atom_vec_meso.h:90) -------------------------------------
void launch_merge_xvt(const AtomSoA &a, float4 *coord4, float4 *veloc4, double cx, double cy, double cz,
                      uint32_t seed, int beg, int end, hipStream_t s);
// (poison, nullable: a device flag - non-zero: the kernel changes nothing; Engine::run redoes the rebuild that raised it)
void launch_nve_initial(const AtomSoA &a, double dtf, double dtv, int groupbit, int n, hipStream_t s, const int *poison = nullptr);
void launch_nve_final(const AtomSoA &a, double dtf, int groupbit, int n, hipStream_t s, const int *poison = nullptr);
// initial_integrate + merge of the locals for the step that follows, one pass (the first step of a run that keeps the table)
void launch_nve_initial_merge(const AtomSoA &a, double dtf, double dtv, int groupbit, int n, float4 *coord4, float4 *veloc4, double cx,
                              double cy, double cz, uint32_t seed, hipStream_t s, const int *poison = nullptr);
// final(step s) + initial(step s+1) [+ merge for step s+1] in one pass
void launch_nve_boundary(const AtomSoA &a, double dtf, double dtv, int groupbit, int n, int merge, float4 *coord4,
                         float4 *veloc4, double cx, double cy, double cz, uint32_t seed_next, hipStream_t s, const int *poison = nullptr);
void launch_sum_mv2(const AtomSoA &a, int groupbit, int n, double *partial, double *result, hipStream_t s);
void launch_pbc(const AtomSoA &a, const double *boxlo, const double *boxhi, const int *periodic, int n,
                hipStream_t s);
void launch_copy_f4(const float4 *src, float4 *dst, size_t n, hipStream_t s);     // bandwidth probe (meso_membw_probe)
void launch_fill_f64(double *p, double val, int n, hipStream_t s);
void launch_fill_i32(int *p, int val, int n, hipStream_t s);
void launch_dtfm_table(const double *mass_type, int ntypes, double dtf, double *dtfm_type, hipStream_t s);
void launch_unpack_mass(const int *type, const double *mass_type, int ntypes, double *mass, int beg, int end,
                        hipStream_t s);
void launch_max_disp2(const AtomSoA &a, const double *xhold, int n, int stride, double *partial, double *result,
                      hipStream_t s);
void launch_copy_hold(const AtomSoA &a, double *xhold, int n, int stride, hipStream_t s);

// ---- reorder (atom_meso.cu:268-314, comm_meso.cu:188-254) ---------------------------------------------
// key = [border][Morton(bin)][Morton(16^3 sub-cell)]; returns number of significant key bits.
int reorder_key_bits(const BinGeom &g);
int reorder_sub_bits(const BinGeom &g);
void launch_reorder_keys(const AtomSoA &a, const BinGeom &g, const double *slab_lo, const double *slab_hi,
                         const int *dim_active, uint32_t *key, int *val, int n, hipStream_t s);
// reorder by counting per extended code (no comparison sort): count -> scan (caller) -> place + order
void launch_reorder_count(const AtomSoA &a, const BinGeom &g, const double *slab_lo, const double *slab_hi, uint32_t *key,
                          int *rank, int *cnt, int n, const double *wrap_lo, const double *wrap_hi, const int *wrap_per,
                          hipStream_t s);      // wrap_lo != null: the periodic wrap (k_pbc) happens here
void launch_reorder_place(const uint32_t *key, const int *rank, const int *estart, const BinGeom &g, int ncodes, int n, int cap,
                          int *placed, int *val_sorted, uint32_t *key_sorted, int *cnt, hipStream_t s);
void launch_ghost_order(const uint32_t *code, const int *rank, const int *gstart, int M, int nghost, int cap, int *placed,
                        int *slotval, uint32_t *code_sorted, int *gslot, int *cnt, const int *nghost_dev /*nullable*/, hipStream_t s);
void launch_permute_atoms(const AtomSoA &src, const AtomSoA &dst, const int *perm_from, int n, int with_f, hipStream_t s);
// ... and the merged float4 pair of the new order in the same pass (k_merge_xvt folded in)
void launch_permute_merge(const AtomSoA &src, const AtomSoA &dst, const int *perm_from, int n, int with_f, float4 *coord4,
                          float4 *veloc4, double cx, double cy, double cz, uint32_t seed, int *inverse /*nullable: old -> new place*/,
                          int *zero /*nullable: per-atom counter cleared for the new order*/, hipStream_t s);
void launch_translate_list(int *list, const int *inverse, int bound, const int *n_dev, const int *n_bulk, int *report, hipStream_t s);
void launch_invert_perm(const int *perm_from, int *perm_to, int n, hipStream_t s);

// ---- halo: border lists + pack (comm_meso.cu:41-186, atom_vec_dpd_atomic_meso.cu:61-244) --------------
// 26 directions d = (dx+1) + 3*(dy+1) + 9*(dz+1), centre (13) unused.
void launch_border_count(const AtomSoA &a, int beg, int end, const double *slab_lo, const double *slab_hi,
                         const int *dim_active, int *chunk_count /*[27][nchunk]*/, int nchunk, hipStream_t s, int *clear = nullptr);
void launch_border_fill(const AtomSoA &a, int beg, int end, const double *slab_lo, const double *slab_hi,
                        const int *dim_active, const int *chunk_offset /*[27][nchunk] exclusive, global*/,
                        int nchunk, int *sendlist, hipStream_t s);
void launch_dir_starts(const int *chunk_offset, int nchunk, int *dir_start, hipStream_t s);
// scan + direction starts + checks + host report in one single-workgroup launch; false (nothing launched) when 27 nchunk + 1 > 64 Ki
bool launch_border_scan(const int *cnt, int *off, int nchunk, int *dir_start, const int *n_bulk, int scan_beg, int bound, int *flags,
                        int *report /* device-visible pinned host memory, 64 ints */, hipStream_t s);
void launch_dir_starts_check(const int *chunk_offset, int nchunk, int *dir_start, const int *n_bulk, int scan_beg, int bound, int *flags,
                             hipStream_t s);
void launch_border_count_code(const int *code, int beg, int end, int *chunk_count, int nchunk, hipStream_t s);
void launch_border_fill_code(const int *code, int beg, int end, const int *chunk_offset, int nchunk, int *list,
                             hipStream_t s);
// border payload: fp64 x (+shift), tag, type, mask -> destination arrays (tail of SoA or a send buffer)
void launch_pack_border(const AtomSoA &a, const int *sendlist, int nsend, const int *dir_start /*dev [28]*/,
                        const double *shift27 /*host [27][3]*/, double *dx, double *dy, double *dz, int *dtag,
                        int *dtype, int *dmask, hipStream_t s);
// per-step payload: merged float4 pair in the receiver's frame (centre c_recv), signature included
void launch_pack_forward(const AtomSoA &a, const int *sendlist, int nsend, const int *dir_start /*dev [28]*/,
                         const double *shift27 /*host [27][3]*/, const double *center27 /*host [27][3]*/,
                         uint32_t seed, float4 *dcoord, float4 *dveloc, const int *dest_slot /*nullable*/,
                         int *img_cnt /*nullable: also record each source atom's images*/, int *img, int img_base,
                         const unsigned char *dirs /*nullable: direction per entry instead of the 27 segments*/, hipStream_t s);



// ---- pair force (pair_dpd_meso.cu:91-205, pair_dpd_fast_meso.cu:91-205) -------------------------------
struct NveArgs {
    double *x[3], *v[3];
    const double *mass;
    const int *mask, *tag, *type;
    const double *mass_type;      // per-type masses [ntypes + 1] for callers that know the atom's type (nve_prefetch); null: per-atom reads
    const double *dtfm_type;      // nullable: dtf / mass per type (launch_dtfm_table), so that no reciprocal sits on the path that ends every wave
    double dtf, dtv;
    int groupbit;
    int merge;
    float4 *coord4_next, *veloc4_next;
    double cx, cy, cz;
    uint32_t seed_next;
    // one rank, small boxes: the atom also writes the merged pair of its periodic images (the per-step ghost refresh without a
    // k_pack_forward launch).  img_cnt[i] images, img[8 i + m] = destination index | direction << 26; null: disabled
    const int *img_cnt, *img;
    const int *img_first;         // nullable, one rank: atoms in front of *img_first (the bulk section of the order) have no images - img_cnt is not read for them
    const double *img_shift;      // [27][3] period shifts
    // where the images go: the merged arrays of the next step (one rank, dest = ghost index) or - several ranks - the send staging of
    // the per-step ghost refresh (dest = slot of the coordinates, the velocities img_vofs[d] slots behind; img_center: the receiver's
    // centre per direction [27][3], null: this rank's own) - what k_pack_forward_multi would pack, without its launch
    float4 *img_c4, *img_v4;
    const int *img_vofs;
    const double *img_center;
};
NveArgs make_nve_args(const AtomSoA &a, double dtf, double dtv, int groupbit, int merge, float4 *coord4_next,
                      float4 *veloc4_next, double cx, double cy, double cz, uint32_t seed_next);

// First kernel of the one-rank rebuild (k_fr_count, rebuild.hip) as a per-atom function: the periodic wrap, the atom's extended
// cell code and sub-cell key, its rank inside the code, its (key, index) word into the code's bucket, the tile totals.  The force
// kernel's step-boundary epilogue calls it on the step in front of a rebuild (the new position is in its registers), which spares
// the rebuild its first launch.
struct FrCountArgs {
    double *x[3];
    int *image;
    int wrap;                  // MesoDomain::pbc folded in
    double boxlo[3], boxhi[3];
    int per[3];
    BinGeom g;
    double sl_lo[3], sl_hi[3]; // border slabs
    int sub_bits, M;
    int *cnt, cap;             // counts per extended code [2M+1], bucket capacity
    unsigned long long *bucket, *ovf;
    int *novf, ovf_cap;
    int *ttot;                 // atoms per tile of 64 codes
    int *gttot;                // nullable (one rank): ghosts per tile of 32 ghost cells - every periodic image of a border atom is booked here
    unsigned dir_mask;         // directions that have a receiver
    int *flags;
};

// bonds evaluated by the force kernel's epilogue (nbond null: none)
struct BondArgs {
    const int *nbond, *bond_idx, *bond_type;
    int bpa, nbt, style;
    const double *cf;
    double prd[3];
};
struct PairArgs {
    const float4 *coord4, *veloc4;
    const int *count, *table;
    int n_col;
    double *f[3];
    double *e_pair;       // nullable
    double *virial[6];    // nullable when e_pair is null
    const double *coeff64;
    const float *coeff32;
    int ntypes;
    double cf1[7];        // the single coefficient row when ntypes == 1 (kernel-argument constants, no LDS lookups)
    double dt_inv_sqrt;
    int beg, end;
    int accumulate;       // 1: f += (reference semantics), 0: f = (force_clear fused)
    int chunked;          // 1: chunked-8 rows (cell-ordered builder), 0: transposed 64-atom tiles
    int debug;            // timing ablations only (0 in production): 1 stop after halo copy, 2 skip phase B
    int nall;             // atoms in coord4/veloc4 (locals + ghosts): bound of the buffer-addressed gathers
    int all_expw_one;     // every pair type has weight exponent 1 (no pow() in the kernel)
    int uniform_cut;      // every pair type has the cutoff of cf1 (the cutoff test needs no table)
    const float *ftab;    // dpd/tableforce/meso: [ntypes^2][ftab_len] conservative-force tables over r/rc in [0,1]; null otherwise
    int ftab_len;
    const float *poly;    // dpd/polyforce/meso: [ntypes^2][MESO_POLY_PITCH] conservative-force polynomials; null otherwise
    int rng;              // fp32 styles: 0 TEA Gaussian (dpd/fast/meso), 1 logistic map (dpd/mini/meso), 2 TEA uniform (tableforce)
    int npart;            // ring kernel: lanes per atom (0: chosen from the launch size; 1, 2, 4)
    int lds_veloc;        // ring kernel (set by its launcher): in-group partners' velocity records from the workgroup's LDS copy
    int share;            // ring kernel: Newton pairing inside a workgroup allowed (end == nlocal or a multiple of 256)
    // ring kernel, scheduling hint only: atoms from bulk_hint on are border atoms (more one-sided pairs, periodic images to write).  The
    // launcher deals the bulk workgroups AND the border workgroups out over the eight XCDs (xcd_sb, xcd_sr: shares per XCD; xcd_kb: the
    // first border workgroup): with one contiguous range per XCD the last XCD would hold every border atom.  0 / -1: one range per XCD
    int bulk_hint, xcd_kb, xcd_sb, xcd_sr;
    // partitioned rows (RowPartArgs below): count / table hold the FRONT sections; nback[i] mirrored entries of atom i live in
    // table_back (chunked-8 rows of nb_col entries); part_group = the pairing group the builder partitioned for.  nback null: plain rows
    const int *nback, *table_back;
    int nb_col, part_group;
    // ring kernel epilogue: the step boundary of the atoms this launch owns (fuse_nve != 0; forces are then not stored)
    int fuse_nve;
    NveArgs nve;
    int frc_on;           // with fuse_nve: the epilogue also runs the rebuild's count over the positions it has just written
    FrCountArgs frc;
    BondArgs bond;        // with fuse_nve: this atom's bond forces are computed in the epilogue and added before the step boundary
    // ring kernel, fp32 styles: set to 1 when the 32-bit fixed-point force sum of an atom (16 fractional bits: +-32768 force units)
    // is beyond half its range (nullable).  Engine::check_overflow turns it into an error instead of a silently wrapped force.
    int *range_flag;
    // non-zero: a rebuild of this interval reported an outgrown capacity (ghost list, cell bucket, border message).  The launch then
    // computes but STORES nothing - no step boundary, no forces, no rebuild count - so the state stays what the rebuild left and
    // Engine::run can redo that rebuild through the synchronous path (requested at kernel entry, looked at in front of the epilogue)
    const int *poison;
};
void launch_pair_dpd(const PairArgs &p, int fast, int evflag, hipStream_t s);
// fp32 style on chunked-8 rows: light cutoff scan per lane, hits compacted into a per-wave LDS ring of 4-byte
// records, heavy phase on full waves with the partner data re-gathered through buffer loads (pair_ring.hip)
int pair_ring_group();   // atoms per workgroup of the ring kernel (alignment of paired launches)
int pair_ring_group_for(int n, int npart_opt);   // pairing group of a launch over n atoms (64 / lanes per atom * waves)
// variant_out (nullable, 128 bytes): the instantiation launched, spelled as a profiler prints it
bool launch_pair_dpd_ring(const PairArgs &p, int fast, hipStream_t s, char *variant_out = nullptr);
// cell-ordered list builder (locals in reorder order, ghosts sorted by Morton bin)
void launch_bin_ranges(const int *estart, const int *gstart, int M, int nlocal, int4 *binrange, hipStream_t s);
struct ExclArgs;
void launch_cell_build(const float4 *coord4, const uint32_t *sorted_key, int key_shift, const int4 *binrange, int M,
                       const int *mbin, float rc2, int nlocal, int n_col, int *count, int *table, int *overflow,
                       const ExclArgs *excl, hipStream_t s);

// ---- bricks (brick.hip): Morton-aligned 4x4x4-bin bricks with a per-rebuild plan of their 6x6x6-bin neighbourhoods; used by
// the tile list builder of the cell-ordered layout and by the brick layout (LDS-staged force kernel, 16-bit rows) ---------
struct BrickArgs {
    const int *estart;   // [2M+1] first local index per extended code (border*M + Morton(bin))
    const int *gstart;   // [M+1]  first sorted-ghost slot per Morton(bin)
    const int *gcnt;     // null, or [M] ghosts per cell: then gstart is valid for ghost cells only (fused rebuild, inline plan)
    int ghost_base;      // == nlocal: ghosts live behind the locals in the merged arrays
    int M;               // Morton codes per section (power of 8)
    int mbin[3];
    float org[3], binw[3];   // lower corner of bin (0,0,0) and the bin widths (fp32: only the origin of a brick's relative coordinates)
    int nbricks;         // M / 64
    const int *active;   // [nactive] ids of bricks that own atoms; null: identity (every brick, empty ones exit)
    int maxh;            // halo atoms a brick may hold (pitch of hmap, LDS of the tile builder); chosen from the density
    int maxh2;           // the same for the 2x2x2 brick of small boxes (4x4x4-bin neighbourhood); 0: never use it
    int brick2_limit;    // 2-bricks while the bin grid spans at most this many 4-bricks
    int split2;          // workgroups per 2-brick (0: by the number of bricks)
    const int *order2;   // [norder2] the 2-bricks that own real cells, by decreasing number of them (null: every brick, in Morton order)
    int norder2;
    int maxown;          // atoms a brick may own (0: unlimited - only the brick-layout kernels have a static bound)
    int nactive;         // launch bound: number of active bricks, or of all bricks when the count is only on the device
    const int *nactive_dev;   // null: nactive is exact
    int *queue;          // tile list builder, 2-bricks: null = one workgroup per brick; else tile_build_queue_ints() zeroed ints (workgroups done, brick counters): persistent workgroups
    // written by the plan kernel once per rebuild (pitches: brick_*_pitch())
    int *hoff;           // [slot][217] first halo slot of each halo bin
    uint32_t *hmap;      // [slot][nh]  halo slot -> global atom index
    int *hdr;            // [slot]{nh, o0, n0, o1, n1}: halo size, own runs of the bulk and border sections
    uint32_t *own_info;  // [nlocal]    own atom -> halo slot | halo bin << 16
    int plan_inline;     // tile builder computes the brick's plan itself (no k_brick_plan launch; identity brick list only)
};
int brick_codes();
int brick_static_maxh();
int brick_static_maxown();
int tile_build_maxh_limit(int n_col, int with_tags);   // with_tags: special-bond filter active (tags staged too)
size_t brick_hoff_pitch();
size_t brick_hdr_pitch();
void launch_brick_plan(const BrickArgs &g, int *overflow, hipStream_t s);
// ghost binning by counting (no sort): cnt[M+1] zeroed by the caller, scanned into gstart between the two calls
// nghost_dev != null: nghost is only a launch bound, the count itself is read on the device
void launch_ghost_count(const AtomSoA &a, const BinGeom &g, int nlocal, int nghost, uint32_t *code, int *rank, int *cnt,
                        const int *nghost_dev, hipStream_t s);

struct ExclArgs;
// Partitioned rows (round 5).  The tile builder knows both indices of every kept candidate, so it decides ONCE per rebuild what the
// force kernel decided per entry and step: a pair whose two atoms lie in the same aligned group of `group` atoms (the ring kernel's
// workgroup) is evaluated by exactly one of them and added to both (Newton pairing).  The rule is balanced - atom i evaluates the
// pair (i, j) when (i < j) != ((i ^ j) & 1) - so every atom keeps about half of its in-group partners, whatever its place in the
// group (the round-4 rule "the lower index evaluates" gave a group's first wave 2.4 times the pairs of its last).  A row comes in
// two sections, each padded with the atom itself to whole 32-byte chunks:
//   front (count[i] entries of table):     what this atom evaluates - in-group partners for both atoms, the others one-sided;
//   back  (nback[i] entries of table_back): mirrored entries - in-group partners that evaluate the pair themselves.
// A kernel that needs every neighbour walks both; a pairing launch of the ring kernel walks the front section only.  The reference
// keeps its rows in two sections for a like reason ("core from the row front, skin from the back", neigh_build_meso.cu:91-115).
struct RowPartArgs {
    int group;           // pairing group of the force launches this table serves (power of two); 0: plain rows, nothing mirrored
    int nlocal;          // partners from this index on (ghosts) are never paired
    int *nback, *back;   // [nlocal] out, and the back table (chunked-8 rows of nb_col entries)
    int nb_col;
};
// cell-ordered layout: wave-per-bin ballot builder on the LDS-staged neighbourhood, chunked-8 global-index rows
int tile_build_queue_ints();
void launch_tile_build(const BrickArgs &g, const float4 *coord4, float rc2, int n_col, int *count, int *table, int *overflow,
                       const ExclArgs *excl, int nlocal, int dbg, hipStream_t s, const RowPartArgs *part = nullptr);
int tile_build_rowcap();
void launch_estart(const uint32_t *sorted_key, int n, int key_shift, int ncodes, int *estart, hipStream_t s);
void launch_code_starts_u32(const uint32_t *sorted_key, int n, int ncodes, int *start, hipStream_t s);
void launch_ghost_morton(const AtomSoA &a, const BinGeom &g, int nlocal, int nghost, uint32_t *key, int *val,
                         hipStream_t s);

// ---- bonded topology (bond.hip) -----------------------------------------------------------------------
void launch_tag_cell(const int *tag, const int *gslot, int nlocal, int nghost, const int *nghost_dev, int *tagc, hipStream_t s);
void launch_set_map(const int *tagc, int nlocal, int nghost, const int *nghost_dev, int maxtag, const uint32_t *tagbits, int *map, hipStream_t s);
void launch_map_bonds(const int *nbond, const int *bond_tag, int bpa, const int *map, int maxtag, int nlocal,
                      int *bond_idx, int *missing, hipStream_t s);
// gpu_map_angle (neighbor_meso.cu:161-182): tags -> indices, the atom's own tag -> itself
void launch_map_angles(const int *tag, const int *nangle, const int *angle_tag, int apa, const int *map, int maxtag, int nlocal,
                       int *angle_idx, int *missing, hipStream_t s);
// gpu_angle_harmonic (angle_harmonic_meso.cu:46-172): cf = [k][theta0 (radians)]
void launch_angle_harmonic(const float4 *coord4, const int *nangle, const int *angle_idx, const int *angle_tag, int apa,
                           const double *cf, int nat, const double *prd, int nlocal, double *fx, double *fy, double *fz,
                           double *e_angle, hipStream_t s);
// style 0: harmonic, cf = [k][r0]; style 1: FENE, cf = [k][r0][epsilon][sigma] (each nbt + 1 long)
void launch_bond(int style, const float4 *coord4, const int *nbond, const int *bond_idx, const int *bond_type, int bpa,
                 const double *cf, int nbt, const double *prd, int nlocal, double *fx, double *fy, double *fz, double *e_bond,
                 int store, hipStream_t s);      // store: f = bond force (opens the step's forces) instead of f += bond force
struct ExclArgs {            // special-partner filter of the list builder (null tagc: no exclusions)
    const int *tagc, *nspecial, *special;
    int msp;
};

// ---- unit kernels for known-answer tests --------------------------------------------------------------
void launch_test_tea(const uint32_t *u, const uint32_t *v, int n, int rounds, uint32_t *out0, uint32_t *out1,
                     hipStream_t s);
void launch_test_logistic(const uint32_t *u, const uint32_t *v, int n, float *out, hipStream_t s);
void launch_test_gaussian(const uint32_t *u, const uint32_t *v, int n, double *out_dp, float *out_sp,
                          hipStream_t s);

} // namespace meso
