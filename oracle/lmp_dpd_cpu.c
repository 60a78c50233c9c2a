/* ---------------------------------------------------------------------------
 * ORACLE (test infrastructure, NOT product code).
 *
 * CPU restatement of the reference's *stock* LAMMPS DPD path on one MPI rank:
 *   Verlet::run            /root/reference/src/verlet.cpp:227-314
 *   FixNVE                 /root/reference/src/fix_nve.cpp:64-140
 *   PairDPD::compute       /root/reference/src/pair_dpd.cpp:63-158 (+ init_one :265-277)
 *   RanMars                /root/reference/src/random_mars.cpp:25-116
 *   RanPark                /root/reference/src/random_park.cpp:20-49
 *   Velocity::create       /root/reference/src/velocity.cpp:140-330 (loop all, dist uniform,
 *                          mom yes, rot no), zero_momentum :683-708, rescale :661-677
 *   Neighbor::setup_bins   /root/reference/src/neighbor.cpp:1459-1620
 *   Neighbor::bin_atoms    /root/reference/src/neighbor.cpp:1787-1823, coord2bin :1837-1866
 *   stencil_half_bin_3d_newton  /root/reference/src/neigh_stencil.cpp:144-159
 *   half_bin_newton        /root/reference/src/neigh_half_bin.cpp:247-364
 *   Neighbor::decide       /root/reference/src/neighbor.cpp:1216-1231
 *   Comm::setup/borders/forward_comm/reverse_comm (1 rank, periodic self swaps)
 *                          /root/reference/src/comm.cpp:393-630, 935-1103
 *   Domain::pbc            /root/reference/src/domain.cpp
 *   ComputeTemp / ComputePressure / ComputePE scalars (units lj)
 *
 * Pinned (tests/test_oracle_lmp.py) against outputs of the reference binary
 * recorded in SURVEY.md 8c / BASELINE.md 2 for example/simple/25.data:
 *   step-0 PE/atom, P; sigma=0 temperature trajectory at steps 10..50; half-list
 *   neighbors/atom; Nghost.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * An OpenMP mode (nthreads > 1) is provided for the cpu_baseline measurement
 * only: it mirrors what USER-OMP style threading does to this algorithm
 * (per-thread force arrays + per-thread RanMars(seed+tid)); nthreads == 1 is
 * the exact sequential algorithm and the only mode that is golden-pinned.
 * ------------------------------------------------------------------------- */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define EPSILON 1.0e-10
#define SMALL 1.0e-6
#define BIG 1.0e20

/* ------------------------------ RanMars ---------------------------------- */
typedef struct {
    double u[98];
    int i97, j97;
    double c, cd, cm;
    int save;
    double second;
} RanMars;

static double mars_uniform(RanMars *r)
{
    double uni = r->u[r->i97] - r->u[r->j97];
    if (uni < 0.0) uni += 1.0;
    r->u[r->i97] = uni;
    r->i97--;
    if (r->i97 == 0) r->i97 = 97;
    r->j97--;
    if (r->j97 == 0) r->j97 = 97;
    r->c -= r->cd;
    if (r->c < 0.0) r->c += r->cm;
    uni -= r->c;
    if (uni < 0.0) uni += 1.0;
    return uni;
}

static int mars_init(RanMars *r, int seed)
{
    if (seed <= 0 || seed > 900000000) return -1;
    r->save = 0;
    int ij = (seed - 1) / 30082;
    int kl = (seed - 1) - 30082 * ij;
    int i = (ij / 177) % 177 + 2;
    int j = ij % 177 + 2;
    int k = (kl / 169) % 178 + 1;
    int l = kl % 169;
    for (int ii = 1; ii <= 97; ii++) {
        double s = 0.0, t = 0.5;
        for (int jj = 1; jj <= 24; jj++) {
            int m = ((i * j) % 179) * k % 179;
            i = j; j = k; k = m;
            l = (53 * l + 1) % 169;
            if ((l * m) % 64 >= 32) s = s + t;
            t = 0.5 * t;
        }
        r->u[ii] = s;
    }
    r->c = 362436.0 / 16777216.0;
    r->cd = 7654321.0 / 16777216.0;
    r->cm = 16777213.0 / 16777216.0;
    r->i97 = 97;
    r->j97 = 33;
    mars_uniform(r);
    return 0;
}

static double mars_gaussian(RanMars *r)
{
    double first, v1, v2, rsq, fac;
    if (!r->save) {
        int again = 1;
        while (again) {
            v1 = 2.0 * mars_uniform(r) - 1.0;
            v2 = 2.0 * mars_uniform(r) - 1.0;
            rsq = v1 * v1 + v2 * v2;
            if (rsq < 1.0 && rsq != 0.0) again = 0;
        }
        fac = sqrt(-2.0 * log(rsq) / rsq);
        r->second = v1 * fac;
        first = v2 * fac;
        r->save = 1;
    } else {
        first = r->second;
        r->save = 0;
    }
    return first;
}

/* ------------------------------ RanPark ---------------------------------- */
#define IA 16807
#define IM 2147483647
#define AM (1.0 / IM)
#define IQ 127773
#define IR 2836

static double park_uniform(int *seed)
{
    int k = *seed / IQ;
    *seed = IA * (*seed - k * IQ) - IR * k;
    if (*seed < 0) *seed += IM;
    return AM * (*seed);
}

/* the two generators as plain streams (n uniform() draws, then n gaussian() draws): what tests/test_oracle_ref.py
 * compares bit for bit with the reference's own random_mars.cpp / random_park.cpp built into oracle/_ref */
int lmp_ranmars_stream(int seed, int n, double *uni, double *gau)
{
    RanMars r;
    if (mars_init(&r, seed)) return -1;
    for (int i = 0; i < n; i++) uni[i] = mars_uniform(&r);
    for (int i = 0; i < n; i++) gau[i] = mars_gaussian(&r);
    return 0;
}

/* RanPark::gaussian (random_park.cpp:53-76): the same polar method on the Park-Miller stream */
int lmp_ranpark_stream(int seed, int n, double *uni, double *gau)
{
    if (seed <= 0) return -1;
    int s = seed, save = 0;
    double second = 0.0;
    for (int i = 0; i < n; i++) uni[i] = park_uniform(&s);
    for (int i = 0; i < n; i++) {
        if (!save) {
            double v1, v2, rsq;
            do {
                v1 = 2.0 * park_uniform(&s) - 1.0;
                v2 = 2.0 * park_uniform(&s) - 1.0;
                rsq = v1 * v1 + v2 * v2;
            } while (rsq >= 1.0 || rsq == 0.0);
            double fac = sqrt(-2.0 * log(rsq) / rsq);
            second = v1 * fac;
            gau[i] = v2 * fac;
            save = 1;
        } else {
            gau[i] = second;
            save = 0;
        }
    }
    return 0;
}

/* ------------------------------ system ----------------------------------- */
typedef struct {
    /* atoms */
    int nlocal, nghost, nmax, ntypes;
    double (*x)[3], (*v)[3], (*f)[3];
    int *type, *tag;
    double *mass; /* per type, 1-based */
    /* box */
    double boxlo[3], boxhi[3], prd[3];
    /* pair */
    double temperature, cut_global;
    double *cut, *a0, *gamma, *sigma, *cutsq; /* (ntypes+1)^2 */
    int seed;
    RanMars random;
    RanMars *trandom; /* per-thread RNGs (OpenMP baseline mode) */
    double eng_vdwl, virial[6];
    /* neighbor */
    double skin, cutneighmax, cutneighmaxsq;
    double *cutneighsq;
    int every, delay, ago;
    int nbinx, nbiny, nbinz, mbinx, mbiny, mbinz, mbinxlo, mbinylo, mbinzlo, mbins;
    double binsizex, binsizey, binsizez, bininvx, bininvy, bininvz;
    int *binhead, *bins;
    int maxbins_atoms;
    int nstencil, *stencil;
    int *numneigh;
    long *firstneigh;     /* offsets into neighpool */
    int *neighpool;
    long neighpool_cap;
    long nneigh_total;
    /* comm */
    double cutghost;
    double slablo[6], slabhi[6];
    int pbc[6][3], pbc_flag[6];
    int *sendlist[6];
    int sendnum[6], recvnum[6], firstrecv[6], maxsendlist[6];
    /* integrate */
    double dt;
    long ntimestep;
    int nthreads;
    double (*tf)[3]; /* per-thread force scratch, nthreads * nmax */
    int *twin;       /* per-thread touched index windows [lo,hi) for owned and ghost partners: 4 ints */
    int nbuild;
    long nextsort;
    int sortfreq;
} LmpSys;

static void *xrealloc(void *p, size_t n)
{
    void *q = realloc(p, n ? n : 1);
    if (!q) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
    return q;
}

static void grow_atoms(LmpSys *s, int nmax)
{
    if (nmax <= s->nmax) return;
    nmax = nmax + nmax / 4 + 1024;
    s->x = xrealloc(s->x, sizeof(double[3]) * nmax);
    s->v = xrealloc(s->v, sizeof(double[3]) * nmax);
    s->f = xrealloc(s->f, sizeof(double[3]) * nmax);
    s->type = xrealloc(s->type, sizeof(int) * nmax);
    s->tag = xrealloc(s->tag, sizeof(int) * nmax);
    s->bins = xrealloc(s->bins, sizeof(int) * nmax);
    if (s->nthreads > 1) s->tf = xrealloc(s->tf, sizeof(double[3]) * (size_t)nmax * s->nthreads);
    s->nmax = nmax;
}

LmpSys *lmp_create(int natoms, int ntypes, const double *boxlo, const double *boxhi,
                   const double *x /* [n][3] */, const int *type, int nthreads)
{
    LmpSys *s = calloc(1, sizeof(LmpSys));
    s->ntypes = ntypes;
    s->nthreads = nthreads < 1 ? 1 : nthreads;
    for (int d = 0; d < 3; d++) {
        s->boxlo[d] = boxlo[d];
        s->boxhi[d] = boxhi[d];
        s->prd[d] = boxhi[d] - boxlo[d];
    }
    grow_atoms(s, natoms);
    s->nlocal = natoms;
    for (int i = 0; i < natoms; i++) {
        s->x[i][0] = x[3 * i]; s->x[i][1] = x[3 * i + 1]; s->x[i][2] = x[3 * i + 2];
        s->v[i][0] = s->v[i][1] = s->v[i][2] = 0.0;
        s->f[i][0] = s->f[i][1] = s->f[i][2] = 0.0;
        s->type[i] = type ? type[i] : 1;
        s->tag[i] = i + 1;
    }
    int n1 = ntypes + 1;
    s->mass = calloc(n1, sizeof(double));
    for (int t = 1; t <= ntypes; t++) s->mass[t] = 1.0;
    s->cut = calloc(n1 * n1, sizeof(double));
    s->a0 = calloc(n1 * n1, sizeof(double));
    s->gamma = calloc(n1 * n1, sizeof(double));
    s->sigma = calloc(n1 * n1, sizeof(double));
    s->cutsq = calloc(n1 * n1, sizeof(double));
    s->cutneighsq = calloc(n1 * n1, sizeof(double));
    s->skin = 0.3;
    s->every = 1; s->delay = 10; /* LAMMPS defaults; the scripts override */
    s->dt = 0.005;
    s->sortfreq = 1000; /* atom.cpp:62 */
    return s;
}

void lmp_destroy(LmpSys *s)
{
    if (!s) return;
    free(s->x); free(s->v); free(s->f); free(s->type); free(s->tag); free(s->bins);
    free(s->mass); free(s->cut); free(s->a0); free(s->gamma); free(s->sigma); free(s->cutsq);
    free(s->cutneighsq); free(s->binhead); free(s->stencil); free(s->numneigh);
    free(s->firstneigh); free(s->neighpool); free(s->tf); free(s->trandom); free(s->twin);
    for (int i = 0; i < 6; i++) free(s->sendlist[i]);
    free(s);
}

void lmp_set_velocities(LmpSys *s, const double *v)
{
    for (int i = 0; i < s->nlocal; i++)
        for (int d = 0; d < 3; d++) s->v[i][d] = v[3 * i + d];
}

void lmp_set_mass(LmpSys *s, int type, double m) { s->mass[type] = m; }
void lmp_set_timestep(LmpSys *s, double dt) { s->dt = dt; }
/* atom_modify sort N binsize (atom.cpp:62: default 1000; 0 switches Atom::sort off - the setting under which
 * oracle/_ref, which cannot contain atom.cpp, is compared with this file) */
void lmp_set_sortfreq(LmpSys *s, int freq) { s->sortfreq = freq; }
void lmp_set_neighbor(LmpSys *s, double skin, int every, int delay)
{
    s->skin = skin; s->every = every; s->delay = delay;
}

/* pair_style dpd T cut seed  (pair_dpd.cpp:186-208) */
int lmp_pair_style_dpd(LmpSys *s, double T, double cut, int seed)
{
    if (seed <= 0) return -1;
    s->temperature = T; s->cut_global = cut; s->seed = seed;
    if (mars_init(&s->random, seed + 0 /* comm->me */)) return -1;
    if (s->nthreads > 1) {
        s->trandom = xrealloc(s->trandom, sizeof(RanMars) * s->nthreads);
        for (int t = 0; t < s->nthreads; t++) mars_init(&s->trandom[t], seed + t);
    }
    return 0;
}

/* pair_coeff i j a0 gamma [cut]  (pair_dpd.cpp:214-241) + init_one (:265-277) */
void lmp_pair_coeff(LmpSys *s, int i, int j, double a0, double gamma, double cut)
{
    int n1 = s->ntypes + 1;
    if (cut <= 0.0) cut = s->cut_global;
    double sigma = sqrt(2.0 * 1.0 /* boltz, lj */ * s->temperature * gamma);
    int ij = i * n1 + j, ji = j * n1 + i;
    s->a0[ij] = s->a0[ji] = a0;
    s->gamma[ij] = s->gamma[ji] = gamma;
    s->cut[ij] = s->cut[ji] = cut;
    s->sigma[ij] = s->sigma[ji] = sigma;
    s->cutsq[ij] = s->cutsq[ji] = cut * cut;
}

/* Velocity::create, loop all / dist uniform / mom yes / rot no */
void lmp_velocity_create(LmpSys *s, double t_desired, int seed)
{
    int n = s->nlocal;
    /* atom->map(i): tag -> local index */
    int *map = malloc(sizeof(int) * (n + 1));
    for (int i = 0; i < n; i++) map[s->tag[i]] = i;
    int rs = seed;
    for (int i = 1; i <= n; i++) {
        double vx = park_uniform(&rs), vy = park_uniform(&rs), vz = park_uniform(&rs);
        int m = map[i];
        double factor = 1.0 / sqrt(s->mass[s->type[m]]);
        s->v[m][0] = vx * factor; s->v[m][1] = vy * factor; s->v[m][2] = vz * factor;
    }
    free(map);
    /* zero_momentum: Group::mass, Group::vcm (group.cpp:864-902) */
    double masstotal = 0.0, p[3] = {0, 0, 0};
    for (int i = 0; i < n; i++) masstotal += s->mass[s->type[i]];
    for (int i = 0; i < n; i++) {
        double massone = s->mass[s->type[i]];
        p[0] += s->v[i][0] * massone; p[1] += s->v[i][1] * massone; p[2] += s->v[i][2] * massone;
    }
    if (masstotal > 0.0) { p[0] /= masstotal; p[1] /= masstotal; p[2] /= masstotal; }
    for (int i = 0; i < n; i++) { s->v[i][0] -= p[0]; s->v[i][1] -= p[1]; s->v[i][2] -= p[2]; }
    /* temperature + rescale */
    double t = 0.0;
    for (int i = 0; i < n; i++)
        t += (s->v[i][0] * s->v[i][0] + s->v[i][1] * s->v[i][1] + s->v[i][2] * s->v[i][2]) *
             s->mass[s->type[i]];
    double dof = 3.0 * n - 3.0;
    t *= 1.0 / dof; /* mvv2e / (dof*boltz) */
    double factor = sqrt(t_desired / t);
    for (int i = 0; i < n; i++) { s->v[i][0] *= factor; s->v[i][1] *= factor; s->v[i][2] *= factor; }
}

/* ------------------------------ domain / comm ---------------------------- */
static void domain_pbc(LmpSys *s)
{
    for (int i = 0; i < s->nlocal; i++)
        for (int d = 0; d < 3; d++) {
            if (s->x[i][d] < s->boxlo[d]) s->x[i][d] += s->prd[d];
            if (s->x[i][d] >= s->boxhi[d]) {
                s->x[i][d] -= s->prd[d];
                if (s->x[i][d] < s->boxlo[d]) s->x[i][d] = s->boxlo[d];
            }
        }
}

static void comm_setup(LmpSys *s)
{
    s->cutghost = s->cutneighmax;
    int iswap = 0;
    for (int dim = 0; dim < 3; dim++) {
        /* maxneed = int(cutghost*1/prd)+1 == 1 required */
        for (int ineed = 0; ineed < 2; ineed++) {
            s->pbc_flag[iswap] = 0;
            s->pbc[iswap][0] = s->pbc[iswap][1] = s->pbc[iswap][2] = 0;
            if (ineed % 2 == 0) {
                s->slablo[iswap] = -BIG;
                s->slabhi[iswap] = s->boxlo[dim] + s->cutghost;
                s->pbc_flag[iswap] = 1; /* myloc == 0 */
                s->pbc[iswap][dim] = 1;
            } else {
                s->slablo[iswap] = s->boxhi[dim] - s->cutghost;
                s->slabhi[iswap] = BIG;
                s->pbc_flag[iswap] = 1; /* myloc == procgrid-1 */
                s->pbc[iswap][dim] = -1;
            }
            iswap++;
        }
    }
}

static void comm_borders(LmpSys *s)
{
    s->nghost = 0;
    int iswap = 0;
    for (int dim = 0; dim < 3; dim++) {
        int nfirst = 0, nlast = 0;
        for (int ineed = 0; ineed < 2; ineed++) {
            double lo = s->slablo[iswap], hi = s->slabhi[iswap];
            if (ineed % 2 == 0) { nfirst = nlast; nlast = s->nlocal + s->nghost; }
            int nsend = 0;
            for (int i = nfirst; i < nlast; i++)
                if (s->x[i][dim] >= lo && s->x[i][dim] <= hi) {
                    if (nsend == s->maxsendlist[iswap]) {
                        s->maxsendlist[iswap] = s->maxsendlist[iswap] * 3 / 2 + 1024;
                        s->sendlist[iswap] = xrealloc(s->sendlist[iswap], sizeof(int) * s->maxsendlist[iswap]);
                    }
                    s->sendlist[iswap][nsend++] = i;
                }
            /* pack_border_vel + self copy + unpack_border_vel */
            int first = s->nlocal + s->nghost;
            grow_atoms(s, first + nsend);
            double dx = s->pbc[iswap][0] * s->prd[0], dy = s->pbc[iswap][1] * s->prd[1],
                   dz = s->pbc[iswap][2] * s->prd[2];
            for (int k = 0; k < nsend; k++) {
                int j = s->sendlist[iswap][k], g = first + k;
                s->x[g][0] = s->x[j][0] + dx; s->x[g][1] = s->x[j][1] + dy; s->x[g][2] = s->x[j][2] + dz;
                s->v[g][0] = s->v[j][0]; s->v[g][1] = s->v[j][1]; s->v[g][2] = s->v[j][2];
                s->type[g] = s->type[j]; s->tag[g] = s->tag[j];
            }
            s->sendnum[iswap] = s->recvnum[iswap] = nsend;
            s->firstrecv[iswap] = first;
            s->nghost += nsend;
            iswap++;
        }
    }
}

static void comm_forward(LmpSys *s)
{
    for (int iswap = 0; iswap < 6; iswap++) {
        double dx = s->pbc[iswap][0] * s->prd[0], dy = s->pbc[iswap][1] * s->prd[1],
               dz = s->pbc[iswap][2] * s->prd[2];
        int first = s->firstrecv[iswap];
        for (int k = 0; k < s->sendnum[iswap]; k++) {
            int j = s->sendlist[iswap][k], g = first + k;
            s->x[g][0] = s->x[j][0] + dx; s->x[g][1] = s->x[j][1] + dy; s->x[g][2] = s->x[j][2] + dz;
            s->v[g][0] = s->v[j][0]; s->v[g][1] = s->v[j][1]; s->v[g][2] = s->v[j][2];
        }
    }
}

static void comm_reverse(LmpSys *s)
{
    for (int iswap = 5; iswap >= 0; iswap--) {
        int first = s->firstrecv[iswap];
        for (int k = 0; k < s->sendnum[iswap]; k++) {
            int j = s->sendlist[iswap][k], g = first + k;
            s->f[j][0] += s->f[g][0]; s->f[j][1] += s->f[g][1]; s->f[j][2] += s->f[g][2];
        }
    }
}

/* ------------------------------ neighbor --------------------------------- */
static void neigh_init(LmpSys *s)
{
    int n1 = s->ntypes + 1;
    double cutmax = 0.0;
    for (int i = 1; i <= s->ntypes; i++)
        for (int j = 1; j <= s->ntypes; j++) {
            double c = s->cut[i * n1 + j];
            s->cutneighsq[i * n1 + j] = (c + s->skin) * (c + s->skin);
            if (c > cutmax) cutmax = c;
        }
    s->cutneighmax = cutmax + s->skin;
    s->cutneighmaxsq = s->cutneighmax * s->cutneighmax;
}

static double bin_distance(LmpSys *s, int i, int j, int k)
{
    double delx, dely, delz;
    if (i > 0) delx = (i - 1) * s->binsizex; else if (i == 0) delx = 0.0; else delx = (i + 1) * s->binsizex;
    if (j > 0) dely = (j - 1) * s->binsizey; else if (j == 0) dely = 0.0; else dely = (j + 1) * s->binsizey;
    if (k > 0) delz = (k - 1) * s->binsizez; else if (k == 0) delz = 0.0; else delz = (k + 1) * s->binsizez;
    return delx * delx + dely * dely + delz * delz;
}

static void neigh_setup_bins(LmpSys *s)
{
    double bbox[3], bsublo[3], bsubhi[3];
    for (int d = 0; d < 3; d++) {
        bsublo[d] = s->boxlo[d] - s->cutghost;
        bsubhi[d] = s->boxhi[d] + s->cutghost;
        bbox[d] = s->boxhi[d] - s->boxlo[d];
    }
    double binsize_optimal = 0.5 * s->cutneighmax;
    double binsizeinv = 1.0 / binsize_optimal;
    s->nbinx = (int)(bbox[0] * binsizeinv);
    s->nbiny = (int)(bbox[1] * binsizeinv);
    s->nbinz = (int)(bbox[2] * binsizeinv);
    if (s->nbinx == 0) s->nbinx = 1;
    if (s->nbiny == 0) s->nbiny = 1;
    if (s->nbinz == 0) s->nbinz = 1;
    s->binsizex = bbox[0] / s->nbinx; s->binsizey = bbox[1] / s->nbiny; s->binsizez = bbox[2] / s->nbinz;
    s->bininvx = 1.0 / s->binsizex; s->bininvy = 1.0 / s->binsizey; s->bininvz = 1.0 / s->binsizez;

    int mbinxhi, mbinyhi, mbinzhi;
    double coord;
    coord = bsublo[0] - SMALL * bbox[0];
    s->mbinxlo = (int)((coord - s->boxlo[0]) * s->bininvx);
    if (coord < s->boxlo[0]) s->mbinxlo = s->mbinxlo - 1;
    coord = bsubhi[0] + SMALL * bbox[0];
    mbinxhi = (int)((coord - s->boxlo[0]) * s->bininvx);
    coord = bsublo[1] - SMALL * bbox[1];
    s->mbinylo = (int)((coord - s->boxlo[1]) * s->bininvy);
    if (coord < s->boxlo[1]) s->mbinylo = s->mbinylo - 1;
    coord = bsubhi[1] + SMALL * bbox[1];
    mbinyhi = (int)((coord - s->boxlo[1]) * s->bininvy);
    coord = bsublo[2] - SMALL * bbox[2];
    s->mbinzlo = (int)((coord - s->boxlo[2]) * s->bininvz);
    if (coord < s->boxlo[2]) s->mbinzlo = s->mbinzlo - 1;
    coord = bsubhi[2] + SMALL * bbox[2];
    mbinzhi = (int)((coord - s->boxlo[2]) * s->bininvz);

    s->mbinxlo -= 1; mbinxhi += 1; s->mbinx = mbinxhi - s->mbinxlo + 1;
    s->mbinylo -= 1; mbinyhi += 1; s->mbiny = mbinyhi - s->mbinylo + 1;
    s->mbinzlo -= 1; mbinzhi += 1; s->mbinz = mbinzhi - s->mbinzlo + 1;
    s->mbins = s->mbinx * s->mbiny * s->mbinz;
    s->binhead = xrealloc(s->binhead, sizeof(int) * s->mbins);

    int sx = (int)(s->cutneighmax * s->bininvx); if (sx * s->binsizex < s->cutneighmax) sx++;
    int sy = (int)(s->cutneighmax * s->bininvy); if (sy * s->binsizey < s->cutneighmax) sy++;
    int sz = (int)(s->cutneighmax * s->bininvz); if (sz * s->binsizez < s->cutneighmax) sz++;
    int smax = (2 * sx + 1) * (2 * sy + 1) * (2 * sz + 1);
    s->stencil = xrealloc(s->stencil, sizeof(int) * smax);
    int n = 0;
    for (int k = 0; k <= sz; k++)
        for (int j = -sy; j <= sy; j++)
            for (int i = -sx; i <= sx; i++)
                if (k > 0 || j > 0 || (j == 0 && i > 0))
                    if (bin_distance(s, i, j, k) < s->cutneighmaxsq)
                        s->stencil[n++] = k * s->mbiny * s->mbinx + j * s->mbinx + i;
    s->nstencil = n;
}

static inline int coord2bin(const LmpSys *s, const double *x)
{
    int ix, iy, iz;
    if (x[0] >= s->boxhi[0]) ix = (int)((x[0] - s->boxhi[0]) * s->bininvx) + s->nbinx;
    else if (x[0] >= s->boxlo[0]) { ix = (int)((x[0] - s->boxlo[0]) * s->bininvx); if (ix > s->nbinx - 1) ix = s->nbinx - 1; }
    else ix = (int)((x[0] - s->boxlo[0]) * s->bininvx) - 1;
    if (x[1] >= s->boxhi[1]) iy = (int)((x[1] - s->boxhi[1]) * s->bininvy) + s->nbiny;
    else if (x[1] >= s->boxlo[1]) { iy = (int)((x[1] - s->boxlo[1]) * s->bininvy); if (iy > s->nbiny - 1) iy = s->nbiny - 1; }
    else iy = (int)((x[1] - s->boxlo[1]) * s->bininvy) - 1;
    if (x[2] >= s->boxhi[2]) iz = (int)((x[2] - s->boxhi[2]) * s->bininvz) + s->nbinz;
    else if (x[2] >= s->boxlo[2]) { iz = (int)((x[2] - s->boxlo[2]) * s->bininvz); if (iz > s->nbinz - 1) iz = s->nbinz - 1; }
    else iz = (int)((x[2] - s->boxlo[2]) * s->bininvz) - 1;
    return (iz - s->mbinzlo) * s->mbiny * s->mbinx + (iy - s->mbinylo) * s->mbinx + (ix - s->mbinxlo);
}

static void neigh_build(LmpSys *s)
{
    int nlocal = s->nlocal, nall = s->nlocal + s->nghost;
    int n1 = s->ntypes + 1;
    s->ago = 0;
    s->nbuild++;
    for (int i = 0; i < s->mbins; i++) s->binhead[i] = -1;
    for (int i = nall - 1; i >= 0; i--) {
        int ibin = coord2bin(s, s->x[i]);
        s->bins[i] = s->binhead[ibin];
        s->binhead[ibin] = i;
    }
    s->numneigh = xrealloc(s->numneigh, sizeof(int) * (nlocal + 1));
    s->firstneigh = xrealloc(s->firstneigh, sizeof(long) * (nlocal + 1));

    if (s->nthreads == 1) {
        long n = 0;
        for (int i = 0; i < nlocal; i++) {
            if (n + 4096 > s->neighpool_cap) {
                s->neighpool_cap = s->neighpool_cap * 3 / 2 + 64L * nlocal + 8192;
                s->neighpool = xrealloc(s->neighpool, sizeof(int) * s->neighpool_cap);
            }
            int *neighptr = s->neighpool + n;
            int nn = 0;
            int itype = s->type[i];
            double xtmp = s->x[i][0], ytmp = s->x[i][1], ztmp = s->x[i][2];
            for (int j = s->bins[i]; j >= 0; j = s->bins[j]) {
                if (j >= nlocal) {
                    if (s->x[j][2] < ztmp) continue;
                    if (s->x[j][2] == ztmp) {
                        if (s->x[j][1] < ytmp) continue;
                        if (s->x[j][1] == ytmp && s->x[j][0] < xtmp) continue;
                    }
                }
                double delx = xtmp - s->x[j][0], dely = ytmp - s->x[j][1], delz = ztmp - s->x[j][2];
                double rsq = delx * delx + dely * dely + delz * delz;
                if (rsq <= s->cutneighsq[itype * n1 + s->type[j]]) neighptr[nn++] = j;
            }
            int ibin = coord2bin(s, s->x[i]);
            for (int k = 0; k < s->nstencil; k++)
                for (int j = s->binhead[ibin + s->stencil[k]]; j >= 0; j = s->bins[j]) {
                    double delx = xtmp - s->x[j][0], dely = ytmp - s->x[j][1], delz = ztmp - s->x[j][2];
                    double rsq = delx * delx + dely * dely + delz * delz;
                    if (rsq <= s->cutneighsq[itype * n1 + s->type[j]]) neighptr[nn++] = j;
                }
            s->firstneigh[i] = n;
            s->numneigh[i] = nn;
            n += nn;
        }
        s->nneigh_total = n;
    } else {
        /* two-pass (count, fill) so threads can write disjoint ranges; same lists, same order */
#pragma omp parallel for schedule(static) num_threads(s->nthreads)
        for (int i = 0; i < nlocal; i++) {
            int nn = 0, itype = s->type[i];
            double xtmp = s->x[i][0], ytmp = s->x[i][1], ztmp = s->x[i][2];
            for (int j = s->bins[i]; j >= 0; j = s->bins[j]) {
                if (j >= nlocal) {
                    if (s->x[j][2] < ztmp) continue;
                    if (s->x[j][2] == ztmp) {
                        if (s->x[j][1] < ytmp) continue;
                        if (s->x[j][1] == ytmp && s->x[j][0] < xtmp) continue;
                    }
                }
                double delx = xtmp - s->x[j][0], dely = ytmp - s->x[j][1], delz = ztmp - s->x[j][2];
                if (delx * delx + dely * dely + delz * delz <= s->cutneighsq[itype * n1 + s->type[j]]) nn++;
            }
            int ibin = coord2bin(s, s->x[i]);
            for (int k = 0; k < s->nstencil; k++)
                for (int j = s->binhead[ibin + s->stencil[k]]; j >= 0; j = s->bins[j]) {
                    double delx = xtmp - s->x[j][0], dely = ytmp - s->x[j][1], delz = ztmp - s->x[j][2];
                    if (delx * delx + dely * dely + delz * delz <= s->cutneighsq[itype * n1 + s->type[j]]) nn++;
                }
            s->numneigh[i] = nn;
        }
        long n = 0;
        for (int i = 0; i < nlocal; i++) { s->firstneigh[i] = n; n += s->numneigh[i]; }
        s->nneigh_total = n;
        if (n > s->neighpool_cap) {
            s->neighpool_cap = n + n / 8;
            s->neighpool = xrealloc(s->neighpool, sizeof(int) * s->neighpool_cap);
        }
#pragma omp parallel for schedule(static) num_threads(s->nthreads)
        for (int i = 0; i < nlocal; i++) {
            int *neighptr = s->neighpool + s->firstneigh[i];
            int nn = 0, itype = s->type[i];
            double xtmp = s->x[i][0], ytmp = s->x[i][1], ztmp = s->x[i][2];
            for (int j = s->bins[i]; j >= 0; j = s->bins[j]) {
                if (j >= nlocal) {
                    if (s->x[j][2] < ztmp) continue;
                    if (s->x[j][2] == ztmp) {
                        if (s->x[j][1] < ytmp) continue;
                        if (s->x[j][1] == ytmp && s->x[j][0] < xtmp) continue;
                    }
                }
                double delx = xtmp - s->x[j][0], dely = ytmp - s->x[j][1], delz = ztmp - s->x[j][2];
                if (delx * delx + dely * dely + delz * delz <= s->cutneighsq[itype * n1 + s->type[j]]) neighptr[nn++] = j;
            }
            int ibin = coord2bin(s, s->x[i]);
            for (int k = 0; k < s->nstencil; k++)
                for (int j = s->binhead[ibin + s->stencil[k]]; j >= 0; j = s->bins[j]) {
                    double delx = xtmp - s->x[j][0], dely = ytmp - s->x[j][1], delz = ztmp - s->x[j][2];
                    if (delx * delx + dely * dely + delz * delz <= s->cutneighsq[itype * n1 + s->type[j]]) neighptr[nn++] = j;
                }
        }
    }
}

/* ------------------------------ pair dpd --------------------------------- */
static void pair_compute_range(LmpSys *s, int ibeg, int iend, double (*f)[3], RanMars *rng,
                               int evflag, double *eng, double *vir)
{
    int n1 = s->ntypes + 1;
    double dtinvsqrt = 1.0 / sqrt(s->dt);
    for (int i = ibeg; i < iend; i++) {
        double xtmp = s->x[i][0], ytmp = s->x[i][1], ztmp = s->x[i][2];
        double vxtmp = s->v[i][0], vytmp = s->v[i][1], vztmp = s->v[i][2];
        int itype = s->type[i];
        const int *jlist = s->neighpool + s->firstneigh[i];
        int jnum = s->numneigh[i];
        for (int jj = 0; jj < jnum; jj++) {
            int j = jlist[jj];
            double delx = xtmp - s->x[j][0], dely = ytmp - s->x[j][1], delz = ztmp - s->x[j][2];
            double rsq = delx * delx + dely * dely + delz * delz;
            int jtype = s->type[j];
            int ij = itype * n1 + jtype;
            if (rsq < s->cutsq[ij]) {
                double r = sqrt(rsq);
                if (r < EPSILON) continue;
                double rinv = 1.0 / r;
                double delvx = vxtmp - s->v[j][0], delvy = vytmp - s->v[j][1], delvz = vztmp - s->v[j][2];
                double dot = delx * delvx + dely * delvy + delz * delvz;
                double wd = 1.0 - r / s->cut[ij];
                double randnum = mars_gaussian(rng);
                double fpair = s->a0[ij] * wd;
                fpair -= s->gamma[ij] * wd * wd * dot * rinv;
                fpair += s->sigma[ij] * wd * randnum * dtinvsqrt;
                fpair *= rinv; /* factor_dpd == 1 (no special bonds) */
                f[i][0] += delx * fpair; f[i][1] += dely * fpair; f[i][2] += delz * fpair;
                /* newton_pair on: always apply to j */
                f[j][0] -= delx * fpair; f[j][1] -= dely * fpair; f[j][2] -= delz * fpair;
                if (evflag) {
                    *eng += 0.5 * s->a0[ij] * s->cut[ij] * wd * wd;
                    vir[0] += delx * delx * fpair; vir[1] += dely * dely * fpair; vir[2] += delz * delz * fpair;
                    vir[3] += delx * dely * fpair; vir[4] += delx * delz * fpair; vir[5] += dely * delz * fpair;
                }
            }
        }
    }
}

static void force_compute(LmpSys *s, int evflag)
{
    int nall = s->nlocal + s->nghost;
    /* force_clear: newton on -> clear ghosts too */
    memset(s->f, 0, sizeof(double[3]) * nall);
    s->eng_vdwl = 0.0;
    memset(s->virial, 0, sizeof(s->virial));
    if (s->nthreads == 1) {
        pair_compute_range(s, 0, s->nlocal, s->f, &s->random, evflag, &s->eng_vdwl, s->virial);
    } else {
#ifdef _OPENMP
        double eng = 0.0, v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0, v5 = 0;
        if (!s->twin) s->twin = calloc(4 * (size_t)s->nthreads, sizeof(int));
#pragma omp parallel num_threads(s->nthreads) reduction(+ : eng, v0, v1, v2, v3, v4, v5)
        {
            int tid = omp_get_thread_num(), nt = omp_get_num_threads();
            double(*tf)[3] = s->tf + (size_t)tid * s->nmax;
            int chunk = (s->nlocal + nt - 1) / nt;
            int ibeg = tid * chunk, iend = ibeg + chunk;
            if (iend > s->nlocal) iend = s->nlocal;
            if (ibeg > s->nlocal) ibeg = s->nlocal;
            /* atoms are spatially sorted (Atom::sort), so the partners of my block live in a narrow index
             * window plus a slice of the ghosts: only that window of the private array is cleared/reduced
             * (USER-OMP clears and reduces whole per-thread arrays, which does not scale past a few threads) */
            int lo = ibeg, hi = iend, glo = nall, ghi = s->nlocal;
            for (int i = ibeg; i < iend; i++) {
                const int *jl = s->neighpool + s->firstneigh[i];
                for (int jj = 0; jj < s->numneigh[i]; jj++) {
                    int j = jl[jj];
                    if (j < s->nlocal) { if (j < lo) lo = j; if (j >= hi) hi = j + 1; }
                    else { if (j < glo) glo = j; if (j >= ghi) ghi = j + 1; }
                }
            }
            if (glo > ghi) glo = ghi;
            int *w = s->twin + 4 * tid;
            w[0] = lo; w[1] = hi; w[2] = glo; w[3] = ghi;
            memset(tf + lo, 0, sizeof(double[3]) * (hi - lo));
            memset(tf + glo, 0, sizeof(double[3]) * (ghi - glo));
            double e = 0.0, vv[6] = {0, 0, 0, 0, 0, 0};
            pair_compute_range(s, ibeg, iend, tf, &s->trandom[tid], evflag, &e, vv);
            eng += e; v0 += vv[0]; v1 += vv[1]; v2 += vv[2]; v3 += vv[3]; v4 += vv[4]; v5 += vv[5];
#pragma omp barrier
            /* reduce: each thread sums, for its slice of all atoms, the private windows that cover it */
            int c2 = (nall + nt - 1) / nt, b2 = tid * c2, e2 = b2 + c2;
            if (e2 > nall) e2 = nall;
            for (int t = 0; t < nt; t++) {
                double(*sf)[3] = s->tf + (size_t)t * s->nmax;
                const int *wt = s->twin + 4 * t;
                for (int part = 0; part < 2; part++) {
                    int a = wt[2 * part] > b2 ? wt[2 * part] : b2;
                    int b = wt[2 * part + 1] < e2 ? wt[2 * part + 1] : e2;
                    for (int i = a; i < b; i++) {
                        s->f[i][0] += sf[i][0]; s->f[i][1] += sf[i][1]; s->f[i][2] += sf[i][2];
                    }
                }
            }
        }
        s->eng_vdwl = eng;
        s->virial[0] = v0; s->virial[1] = v1; s->virial[2] = v2;
        s->virial[3] = v3; s->virial[4] = v4; s->virial[5] = v5;
#endif
    }
    comm_reverse(s);
}

/* Atom::setup_sort_bins + Atom::sort_local (atom.cpp:1206-1371): spatial sort of owned atoms,
 * bins of 1/2 cutneighmax, stable within a bin.  Called from Verlet::setup (verlet.cpp:111) and on
 * rebuild steps once ntimestep >= nextsort (verlet.cpp).  It fixes the order in which pairs draw
 * from the sequential RanMars stream. */
static void atom_sort(LmpSys *s)
{
    if (s->sortfreq <= 0) return;
    s->nextsort = (s->ntimestep / s->sortfreq) * s->sortfreq + s->sortfreq;
    double binsize = 0.5 * s->cutneighmax, bininv = 1.0 / binsize;
    int nb[3];
    double binv[3];
    for (int d = 0; d < 3; d++) {
        nb[d] = (int)((s->boxhi[d] - s->boxlo[d]) * bininv);
        if (nb[d] == 0) nb[d] = 1;
        binv[d] = nb[d] / (s->boxhi[d] - s->boxlo[d]);
    }
    int nbins = nb[0] * nb[1] * nb[2];
    if (nbins == 1) return;
    int n = s->nlocal;
    int *head = malloc(sizeof(int) * (nbins + 1)), *cell = malloc(sizeof(int) * n), *perm = malloc(sizeof(int) * n);
    memset(head, 0, sizeof(int) * (nbins + 1));
    for (int i = 0; i < n; i++) {
        int b[3];
        for (int d = 0; d < 3; d++) {
            b[d] = (int)((s->x[i][d] - s->boxlo[d]) * binv[d]);
            if (b[d] < 0) b[d] = 0;
            if (b[d] > nb[d] - 1) b[d] = nb[d] - 1;
        }
        cell[i] = b[2] * nb[1] * nb[0] + b[1] * nb[0] + b[0];
        head[cell[i] + 1]++;
    }
    for (int b = 0; b < nbins; b++) head[b + 1] += head[b];
    for (int i = 0; i < n; i++) perm[head[cell[i]]++] = i; /* forward order within a bin */
    double(*nx)[3] = malloc(sizeof(double[3]) * n), (*nv)[3] = malloc(sizeof(double[3]) * n),
    (*nf)[3] = malloc(sizeof(double[3]) * n);
    int *nt = malloc(sizeof(int) * n), *ng = malloc(sizeof(int) * n);
    for (int i = 0; i < n; i++) {
        int j = perm[i];
        memcpy(nx[i], s->x[j], sizeof(double[3])); memcpy(nv[i], s->v[j], sizeof(double[3]));
        memcpy(nf[i], s->f[j], sizeof(double[3]));
        nt[i] = s->type[j]; ng[i] = s->tag[j];
    }
    memcpy(s->x, nx, sizeof(double[3]) * n); memcpy(s->v, nv, sizeof(double[3]) * n);
    memcpy(s->f, nf, sizeof(double[3]) * n);
    memcpy(s->type, nt, sizeof(int) * n); memcpy(s->tag, ng, sizeof(int) * n);
    free(head); free(cell); free(perm); free(nx); free(nv); free(nf); free(nt); free(ng);
}

/* ------------------------------ integrate -------------------------------- */
void lmp_setup(LmpSys *s)
{
    neigh_init(s);
    domain_pbc(s);
    comm_setup(s);
    neigh_setup_bins(s);
    /* exchange: 1 rank no-op */
    atom_sort(s);
    comm_borders(s);
    neigh_build(s);
    s->nbuild = 0;
    force_compute(s, 1);
}

static void nve_initial(LmpSys *s)
{
    double dtv = s->dt, dtf = 0.5 * s->dt;
#pragma omp parallel for schedule(static) num_threads(s->nthreads) if (s->nthreads > 1)
    for (int i = 0; i < s->nlocal; i++) {
        double dtfm = dtf / s->mass[s->type[i]];
        s->v[i][0] += dtfm * s->f[i][0]; s->v[i][1] += dtfm * s->f[i][1]; s->v[i][2] += dtfm * s->f[i][2];
        s->x[i][0] += dtv * s->v[i][0]; s->x[i][1] += dtv * s->v[i][1]; s->x[i][2] += dtv * s->v[i][2];
    }
}

static void nve_final(LmpSys *s)
{
    double dtf = 0.5 * s->dt;
#pragma omp parallel for schedule(static) num_threads(s->nthreads) if (s->nthreads > 1)
    for (int i = 0; i < s->nlocal; i++) {
        double dtfm = dtf / s->mass[s->type[i]];
        s->v[i][0] += dtfm * s->f[i][0]; s->v[i][1] += dtfm * s->f[i][1]; s->v[i][2] += dtfm * s->f[i][2];
    }
}

/* evflag_last: compute energy/virial on the last step (thermo output step) */
void lmp_run(LmpSys *s, int nsteps, int evflag_last)
{
    for (int it = 0; it < nsteps; it++) {
        s->ntimestep++;
        nve_initial(s);
        s->ago++;
        int nflag = (s->ago >= s->delay && s->ago % s->every == 0);
        if (!nflag) {
            comm_forward(s);
        } else {
            domain_pbc(s);
            if (s->sortfreq > 0 && s->ntimestep >= s->nextsort) atom_sort(s);
            comm_borders(s);
            neigh_build(s);
        }
        force_compute(s, evflag_last && it == nsteps - 1);
        nve_final(s);
    }
}

/* ------------------------------ thermo ----------------------------------- */
double lmp_temperature(const LmpSys *s)
{
    double t = 0.0;
    for (int i = 0; i < s->nlocal; i++)
        t += (s->v[i][0] * s->v[i][0] + s->v[i][1] * s->v[i][1] + s->v[i][2] * s->v[i][2]) *
             s->mass[s->type[i]];
    double dof = 3.0 * s->nlocal - 3.0;
    return t / dof;
}

double lmp_pe_per_atom(const LmpSys *s) { return s->eng_vdwl / s->nlocal; }

double lmp_pressure(const LmpSys *s)
{
    double dof = 3.0 * s->nlocal - 3.0;
    double vol = s->prd[0] * s->prd[1] * s->prd[2];
    return (dof * lmp_temperature(s) + s->virial[0] + s->virial[1] + s->virial[2]) / 3.0 / vol;
}

/* eng_vdwl and the six virial components of the last ev step (Pair::eng_vdwl, Pair::virial) */
void lmp_get_ev(const LmpSys *s, double *out7)
{
    out7[0] = s->eng_vdwl;
    for (int k = 0; k < 6; k++) out7[1 + k] = s->virial[k];
}

int lmp_nlocal(const LmpSys *s) { return s->nlocal; }
int lmp_nghost(const LmpSys *s) { return s->nghost; }
long lmp_nneigh(const LmpSys *s) { return s->nneigh_total; }
int lmp_nbuild(const LmpSys *s) { return s->nbuild; }

/* copy out per-atom state ordered by tag (1..N) */
void lmp_get_state(const LmpSys *s, double *x, double *v, double *f)
{
    for (int i = 0; i < s->nlocal; i++) {
        int t = s->tag[i] - 1;
        for (int d = 0; d < 3; d++) {
            if (x) x[3 * t + d] = s->x[i][d];
            if (v) v[3 * t + d] = s->v[i][d];
            if (f) f[3 * t + d] = s->f[i][d];
        }
    }
}
