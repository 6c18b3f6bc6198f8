/*
 * nbody_oracle.c — CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  Nothing under parallelnbody_amd/ links, imports or calls it.
 *
 * PARITY UNPINNED.  The reference (Milias/ParallelNbody) ships no tests, golden vectors or
 * fixtures for this path, and its sources need Unreal Engine 4.9 headers plus
 * UnrealHeaderTool-generated code that this image does not have, so the reference cannot be
 * built here without writing stand-ins for them (not allowed).  This file is therefore a
 * restatement by reading, checked only against analytic known answers and against an
 * independent fp64 direct sum (tests/test_oracle.py).
 *
 * Every function cites the reference lines it restates (paths relative to
 * /root/reference/Source/NBody/).  Arithmetic is restated operation by operation in the
 * reference's types: compile with -ffp-contract=off and without -ffast-math (see Makefile) so
 * that fp32 multiplies and adds stay separate, as FVector's component-wise operators do.
 *
 * Third-party arithmetic that is not under /root/reference: Unreal Engine 4.9 FVector
 * (component-wise IEEE fp32 operators; FVector::Dist = sqrtf of the fp32 sum of squared
 * differences in X,Y,Z order; operator/=(float) multiplies by the fp32 reciprocal;
 * GetAbsMax = max(max(|X|,|Y|),|Z|)) and the C runtime's pow.  The .sln names Visual Studio
 * 2013, whose <cmath> may resolve pow(float,int) to a float overload rather than C++11's
 * double promotion; `pow_mode` selects the reading (0 = double pow, the C++11 rule and the
 * default; 1 = powf; 2 = float d*(d*d); 3 = the correctly rounded cube in double, (d*d)*d — d*d is
 * exact for a float d — which a correctly rounded pow would return and glibc's returns for all but a
 * few arguments in a thousand).  The readings differ by a few fp32 ulp per pair at most.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

/* OctreeSearch.h:8-18 — FParticle, 40-byte AoS record. */
typedef struct {
  float Mass;
  float Position[3];
  float Velocity[3];
  float Acceleration[3];
} oracle_particle;

ORACLE_API int oracle_sizeof_particle(void) { return (int)sizeof(oracle_particle); }

/* ------------------------------------------------------------------------------------------
 * The pair law.  OctreeSearch.h:101-104:
 *   float d = FVector::Dist(CenterOfMass, Particle->Position);
 *   if (d == 0) return;
 *   Acceleration += 1e4*TotalMass/pow(d,3) * (CenterOfMass - Particle->Position);
 * `g` is 1e4 in the reference; `eps2` is 0 in the reference (softening is a build-defined
 * extension: d = sqrtf(|r|^2 + eps2)).
 * ---------------------------------------------------------------------------------------- */
static inline float cube_of(float d, int pow_mode, double *as_double) {
  if (pow_mode == 0) { *as_double = pow((double)d, 3.0); return 0.0f; }
  if (pow_mode == 3) { const double dd = (double)d; *as_double = (dd * dd) * dd; return 0.0f; }
  float p = (pow_mode == 1) ? powf(d, 3.0f) : d * (d * d);
  *as_double = (double)p;
  return p;
}

static inline void pair_law_f32(const float pi[3], const float pj[3], float mj, double g,
                                float eps2, int pow_mode, float acc[3]) {
  /* FVector::Dist(V1=CoM, V2=Pos): sqrtf(Square(V2.X-V1.X)+Square(V2.Y-V1.Y)+Square(V2.Z-V1.Z)) */
  float ex = pi[0] - pj[0], ey = pi[1] - pj[1], ez = pi[2] - pj[2];
  float d2 = ex * ex + ey * ey;
  d2 = d2 + ez * ez;
  if (eps2 != 0.0f) d2 = d2 + eps2;
  float d = sqrtf(d2);
  if (d == 0.0f) return;                                   /* OctreeSearch.h:102 */
  double p3;
  cube_of(d, pow_mode, &p3);
  float s = (float)(g * (double)mj / p3);                  /* double expr → float Scale */
  float dx = pj[0] - pi[0], dy = pj[1] - pi[1], dz = pj[2] - pi[2];   /* CoM - Pos */
  acc[0] = acc[0] + s * dx;                                /* FVector += float*FVector */
  acc[1] = acc[1] + s * dy;
  acc[2] = acc[2] + s * dz;
}

/*
 * All-pairs force pass in body-index order: the theta=0 limit of the loop at
 * OctreeSearch.cpp:83-86 (Acceleration = 0; ComputeForces(i)) with the tree walk replaced by
 * j = 0..n-1.  Bodies [i0,i1) are evaluated against all n.  pos is [n][3], acc is [n][3]
 * (rows i0..i1-1 written).  nthreads>1 uses OpenMP over i (each i is still summed serially).
 */
ORACLE_API int oracle_forces_direct_f32(int n, const float *pos, const float *mass, double g,
                                        float eps2, int pow_mode, int i0, int i1, float *acc,
                                        int nthreads) {
  if (n < 0 || i0 < 0 || i1 > n || i0 > i1) return -1;
#ifdef _OPENMP
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(static) num_threads(nthreads)
#endif
  for (int i = i0; i < i1; ++i) {
    float a[3] = {0.0f, 0.0f, 0.0f};                       /* OctreeSearch.cpp:84 */
    for (int j = 0; j < n; ++j) pair_law_f32(&pos[3 * i], &pos[3 * j], mass[j], g, eps2, pow_mode, a);
    acc[3 * i + 0] = a[0]; acc[3 * i + 1] = a[1]; acc[3 * i + 2] = a[2];
  }
  (void)nthreads;
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * Octree restatement.  OctreeSearch.h:21-109 (class Octree), as an index-linked node pool.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int particle;          /* FParticle* Particle  (-1 = NULL)        .h:24 */
  float origin[3];       /* FVector Origin                           .h:25 */
  float size;            /* float Size  (a half-width)               .h:26 */
  float total_mass;      /* float TotalMass                          .h:27 */
  float com[3];          /* FVector CenterOfMass                     .h:28 */
  int child[8];          /* Octree* Children[8]  (-1 = NULL)         .h:30 */
} onode;

typedef struct {
  onode *nodes;
  int count, cap;
  const float *pos, *mass;
  int overflow;
  int max_depth;         /* deepest node an Add has reached (root = 0): test infrastructure, see oracle_last_max_depth */
  int div_mode;          /* reading of FVector::operator/=(float) in ComputeMass, .h:95: 0 = multiply by the fp32 reciprocal
                            (UE4's implementation as remembered — [external], the engine is not vendored), 1 = divide */
} otree;

#define ORACLE_MAX_DEPTH 200

static int node_new(otree *t, const float origin[3], float size) {   /* ctor .h:33-35 */
  if (t->count == t->cap) {
    int ncap = t->cap ? t->cap * 2 : 1024;
    onode *nn = (onode *)realloc(t->nodes, (size_t)ncap * sizeof(onode));
    if (!nn) { t->overflow = 2; return -1; }
    t->nodes = nn; t->cap = ncap;
  }
  onode *nd = &t->nodes[t->count];
  nd->particle = -1;
  memcpy(nd->origin, origin, sizeof(float) * 3);
  nd->size = size;
  nd->total_mass = 0.0f;
  nd->com[0] = nd->com[1] = nd->com[2] = 0.0f;
  for (int i = 0; i < 8; ++i) nd->child[i] = -1;
  return t->count++;
}

static inline int node_is_leaf(const otree *t, int k) { return t->nodes[k].child[0] == -1; }  /* .h:58 */

static inline int node_octant(const otree *t, int k, const float p[3]) {   /* GetOctant .h:50-56 */
  const onode *nd = &t->nodes[k];
  int o = 0;
  if (p[0] >= nd->origin[0]) o |= 4;
  if (p[1] >= nd->origin[1]) o |= 2;
  if (p[2] >= nd->origin[2]) o |= 1;
  return o;
}

static void node_add(otree *t, int k, int particle, int depth) {     /* Add .h:60-81 */
  if (t->overflow) return;
  if (depth > t->max_depth) t->max_depth = depth;
  if (depth > ORACLE_MAX_DEPTH) { t->overflow = 1; return; }         /* reference recurses forever on duplicates */
  if (node_is_leaf(t, k)) {
    if (t->nodes[k].particle == -1) {
      t->nodes[k].particle = particle;
    } else {
      int old = t->nodes[k].particle;
      t->nodes[k].particle = -1;
      for (int i = 0; i < 8; ++i) {
        float center[3];
        float sz = t->nodes[k].size;
        memcpy(center, t->nodes[k].origin, sizeof(center));
        /* center.X += Size * (i & 4 ? 0.5 : -0.5): double product, float += double   .h:71-73 */
        center[0] = (float)((double)center[0] + (double)sz * ((i & 4) ? 0.5 : -0.5));
        center[1] = (float)((double)center[1] + (double)sz * ((i & 2) ? 0.5 : -0.5));
        center[2] = (float)((double)center[2] + (double)sz * ((i & 1) ? 0.5 : -0.5));
        int c = node_new(t, center, (float)(0.5 * (double)sz));       /* .h:74 */
        if (c < 0) return;
        t->nodes[k].child[i] = c;
      }
      node_add(t, t->nodes[k].child[node_octant(t, k, &t->pos[3 * old])], old, depth + 1);        /* .h:77 */
      node_add(t, t->nodes[k].child[node_octant(t, k, &t->pos[3 * particle])], particle, depth + 1); /* .h:78 */
    }
  } else {
    node_add(t, t->nodes[k].child[node_octant(t, k, &t->pos[3 * particle])], particle, depth + 1);  /* .h:80 */
  }
}

static void node_compute_mass(otree *t, int k) {                      /* ComputeMass .h:83-97 */
  if (node_is_leaf(t, k)) {
    int p = t->nodes[k].particle;
    if (p != -1) {
      memcpy(t->nodes[k].com, &t->pos[3 * p], sizeof(float) * 3);
      t->nodes[k].total_mass = t->mass[p];
    }
  } else {
    for (int i = 0; i < 8; ++i) {
      int c = t->nodes[k].child[i];
      node_compute_mass(t, c);
      onode *nd = &t->nodes[k];
      const onode *ch = &t->nodes[c];
      nd->total_mass = nd->total_mass + ch->total_mass;
      nd->com[0] = nd->com[0] + ch->total_mass * ch->com[0];
      nd->com[1] = nd->com[1] + ch->total_mass * ch->com[1];
      nd->com[2] = nd->com[2] + ch->total_mass * ch->com[2];
    }
    onode *nd = &t->nodes[k];
    if (nd->total_mass != 0.0f) {
      if (t->div_mode == 0) {
        float rv = 1.0f / nd->total_mass;          /* UE4 FVector::operator/=(float): reciprocal multiply */
        nd->com[0] *= rv; nd->com[1] *= rv; nd->com[2] *= rv;
      } else {                                     /* the other reading: three divisions */
        nd->com[0] = nd->com[0] / nd->total_mass; nd->com[1] = nd->com[1] / nd->total_mass; nd->com[2] = nd->com[2] / nd->total_mass;
      }
    } else {
      memcpy(nd->com, nd->origin, sizeof(float) * 3);
    }
  }
}

static void node_forces(const otree *t, int k, const float pi[3], float theta, double g,
                        int pow_mode, float acc[3]) {                 /* ComputeForces .h:99-108 */
  const onode *nd = &t->nodes[k];
  int leaf = node_is_leaf(t, k);
  if (leaf && nd->particle == -1) return;                             /* .h:100 */
  float ex = pi[0] - nd->com[0], ey = pi[1] - nd->com[1], ez = pi[2] - nd->com[2];
  float d2 = ex * ex + ey * ey;
  d2 = d2 + ez * ez;
  float d = sqrtf(d2);                                                /* .h:101 */
  if (d == 0.0f) return;                                              /* .h:102 */
  if (nd->size / d < theta || nd->particle != -1) {                   /* .h:103 */
    double p3;
    cube_of(d, pow_mode, &p3);
    float s = (float)(g * (double)nd->total_mass / p3);               /* .h:104 */
    acc[0] = acc[0] + s * (nd->com[0] - pi[0]);
    acc[1] = acc[1] + s * (nd->com[1] - pi[1]);
    acc[2] = acc[2] + s * (nd->com[2] - pi[2]);
  } else if (!leaf) {
    for (int i = 0; i < 8; ++i) node_forces(t, nd->child[i], pi, theta, g, pow_mode, acc);  /* .h:105-107 */
  }
}

/* DrawOctreeBoxes — OctreeSearch.cpp:36-45: depth first, children 0..7; at every occupied leaf the reference draws
 * DrawDebugBox(Origin, (Size, Size, Size)) when ShowOctree and DrawDebugPoint(Particle->Position).  Recorded here: the
 * leaf's (Origin, Size) and its particle, in that order of visit. */
static void node_draw(const otree *t, int k, float *boxes, int *order, int *count) {
  if (k < 0) return;                                                        /* .cpp:38 */
  const onode *nd = &t->nodes[k];
  if (node_is_leaf(t, k) && nd->particle != -1) {                           /* .cpp:39 */
    if (boxes) { memcpy(&boxes[4 * *count], nd->origin, 12); boxes[4 * *count + 3] = nd->size; }   /* .cpp:40 */
    if (order) order[*count] = nd->particle;                                /* .cpp:41 */
    *count += 1;
  } else {
    for (int i = 0; i < 8; ++i) node_draw(t, nd->child[i], boxes, order, count);   /* .cpp:43 */
  }
}

/*
 * CreateOctree — OctreeSearch.cpp:74-89: root = Octree(root_origin, root_size); Add every
 * particle in index order; ComputeMass; then for each i: Acceleration = 0; ComputeForces(i, theta)
 * (the reference passes theta = 1.0; theta = 0 is exact all-pairs in DFS order).
 * root_com_out receives the root's CenterOfMass, which the next frame's CreateOctree uses as its
 * root origin (.cpp:78-79).  node_count_out (optional) receives the number of nodes.
 * leaf_boxes ([n][4], optional) / leaf_order ([n], optional): what DrawOctreeBoxes (.cpp:36-45) would draw on this
 * tree — (Origin, Size) of every occupied leaf and the particle it holds, in depth-first order.
 * div_mode: see otree.  acc may be NULL (tree only).
 * Returns 0, or 1 if insertion exceeded ORACLE_MAX_DEPTH (duplicate positions: the reference
 * would recurse without bound), 2 on allocation failure.
 */
static int g_last_max_depth = 0;
/* The deepest node the LAST oracle_octree_f32 / oracle_tick_aos*_f32 call on this thread's library reached while inserting
 * (root = 0; ORACLE_MAX_DEPTH + 1 when the insertion was cut off).  Two bodies that share their first L octant digits put
 * leaves at depth L + 1: the device path refuses a frame exactly when this is >= 43 (its keys hold 42 levels), and the
 * fuzz tests ask here whether a refusal was due. */
ORACLE_API int oracle_last_max_depth(void) { return g_last_max_depth; }

ORACLE_API int oracle_octree_f32(int n, const float *pos, const float *mass, const float root_origin[3], float root_size,
                                 float theta, double g, int pow_mode, int div_mode, float *acc, float root_com_out[3],
                                 int *node_count_out, float *leaf_boxes, int *leaf_order) {
  otree t;
  memset(&t, 0, sizeof(t));
  t.pos = pos; t.mass = mass; t.div_mode = div_mode;
  int root = node_new(&t, root_origin, root_size);
  if (root < 0) return 2;
  for (int i = 0; i < n && !t.overflow; ++i) node_add(&t, root, i, 0);      /* .cpp:80 */
  g_last_max_depth = t.max_depth;
  if (t.overflow) { int e = t.overflow; free(t.nodes); return e; }
  node_compute_mass(&t, root);                                              /* .cpp:81 */
  /* (every body's walk reads the finished tree and writes its own three floats: the threads change no bit) */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 256) if (acc && n >= 16384)
#endif
  for (int i = 0; i < (acc ? n : 0); ++i) {                                 /* .cpp:83-86 */
    float a[3] = {0.0f, 0.0f, 0.0f};
    node_forces(&t, root, &pos[3 * i], theta, g, pow_mode, a);
    acc[3 * i + 0] = a[0]; acc[3 * i + 1] = a[1]; acc[3 * i + 2] = a[2];
  }
  if (leaf_boxes || leaf_order) { int cnt = 0; node_draw(&t, root, leaf_boxes, leaf_order, &cnt); }
  if (root_com_out) memcpy(root_com_out, t.nodes[root].com, sizeof(float) * 3);
  if (node_count_out) *node_count_out = t.count;
  free(t.nodes);
  return 0;
}

ORACLE_API int oracle_octree_forces_f32(int n, const float *pos, const float *mass,
                                        const float root_origin[3], float root_size, float theta,
                                        double g, int pow_mode, float *acc, float root_com_out[3],
                                        int *node_count_out) {
  return oracle_octree_f32(n, pos, mass, root_origin, root_size, theta, g, pow_mode, 0, acc, root_com_out, node_count_out, 0, 0);
}

/* Integration loop — OctreeSearch.cpp:28-31:  Velocity += dt*Acceleration; Position += dt*Velocity. */
ORACLE_API void oracle_kick_drift_f32(int n, float *pos, float *vel, const float *acc, float dt) {
  for (int i = 0; i < 3 * n; ++i) {
    vel[i] = vel[i] + dt * acc[i];
    pos[i] = pos[i] + dt * vel[i];
  }
}

/* ComputeCubeSize — OctreeSearch.cpp:47-56: Size = max_i GetAbsMax(Position_i). */
ORACLE_API float oracle_bounds_f32(int n, const float *pos) {
  if (n <= 0) return 0.0f;
  float size = fmaxf(fmaxf(fabsf(pos[0]), fabsf(pos[1])), fabsf(pos[2]));
  for (int i = 1; i < n; ++i) {
    float t = fmaxf(fmaxf(fabsf(pos[3 * i]), fabsf(pos[3 * i + 1])), fabsf(pos[3 * i + 2]));
    if (t > size) size = t;
  }
  return size;
}

/*
 * One physics frame — AOctreeSearch::Tick, OctreeSearch.cpp:25-32, on the 40-byte AoS records:
 *   if (PhDeltaTime > 0) { ComputeCubeSize(); CreateOctree(); kick; drift; }
 * `theta` < 0 selects the index-order direct sum instead of the tree walk.  `root_com` is the
 * actor's persistent "previous tree CoM" (zero before the first frame, .cpp:77-78), updated in place.
 * `size_io` mirrors the actor's Size field.
 */
static int tick_aos(int n, oracle_particle *p, float dt, float theta, double g, int pow_mode, int div_mode, float root_com[3],
                    float *size_io);
ORACLE_API int oracle_tick_aos_f32(int n, oracle_particle *p, float dt, float theta, double g,
                                   int pow_mode, float root_com[3], float *size_io) {
  return tick_aos(n, p, dt, theta, g, pow_mode, 0, root_com, size_io);
}
ORACLE_API int oracle_tick_aos2_f32(int n, oracle_particle *p, float dt, float theta, double g, int pow_mode, int div_mode,
                                    float root_com[3], float *size_io) {
  return tick_aos(n, p, dt, theta, g, pow_mode, div_mode, root_com, size_io);
}
static int tick_aos(int n, oracle_particle *p, float dt, float theta, double g, int pow_mode, int div_mode, float root_com[3],
                    float *size_io) {
  if (!(dt > 0.0f)) return 0;                                               /* .cpp:25 */
  if (n <= 0) return 0;
  float *pos = (float *)malloc(sizeof(float) * 3 * (size_t)n);
  float *mass = (float *)malloc(sizeof(float) * (size_t)n);
  float *acc = (float *)malloc(sizeof(float) * 3 * (size_t)n);
  if (!pos || !mass || !acc) { free(pos); free(mass); free(acc); return 2; }
  for (int i = 0; i < n; ++i) { memcpy(&pos[3 * i], p[i].Position, 12); mass[i] = p[i].Mass; }
  *size_io = oracle_bounds_f32(n, pos);                                     /* .cpp:26 */
  int rc;
  if (theta < 0.0f) {
    rc = oracle_forces_direct_f32(n, pos, mass, g, 0.0f, pow_mode, 0, n, acc, 1);
  } else {
    float com[3];
    rc = oracle_octree_f32(n, pos, mass, root_com, *size_io, theta, g, pow_mode, div_mode, acc, com, 0, 0, 0);
    if (rc == 0) memcpy(root_com, com, sizeof(com));
  }
  if (rc == 0) {
    for (int i = 0; i < n; ++i) {
      memcpy(p[i].Acceleration, &acc[3 * i], 12);
      for (int c = 0; c < 3; ++c) {                                         /* .cpp:29-30 */
        p[i].Velocity[c] = p[i].Velocity[c] + dt * p[i].Acceleration[c];
        p[i].Position[c] = p[i].Position[c] + dt * p[i].Velocity[c];
      }
    }
  }
  free(pos); free(mass); free(acc);
  return rc;
}

/* ------------------------------------------------------------------------------------------
 * fp64 restatements (same law and update in double; build-defined — the reference is fp32).
 * Used as the independent check of the fp32 oracle, as the oracle of the fp64 device path, and
 * for energy diagnostics.
 * ---------------------------------------------------------------------------------------- */
ORACLE_API int oracle_forces_direct_f64(int n, const double *pos, const double *mass, double g,
                                        double eps2, int i0, int i1, double *acc, int nthreads) {
  if (n < 0 || i0 < 0 || i1 > n || i0 > i1) return -1;
#ifdef _OPENMP
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(static) num_threads(nthreads)
#endif
  for (int i = i0; i < i1; ++i) {
    double ax = 0, ay = 0, az = 0;
    const double xi = pos[3 * i], yi = pos[3 * i + 1], zi = pos[3 * i + 2];
    for (int j = 0; j < n; ++j) {
      double dx = pos[3 * j] - xi, dy = pos[3 * j + 1] - yi, dz = pos[3 * j + 2] - zi;
      double d2 = dx * dx + dy * dy + dz * dz + eps2;
      if (d2 == 0.0) continue;
      double d = sqrt(d2);
      double s = g * mass[j] / (d2 * d);
      ax += s * dx; ay += s * dy; az += s * dz;
    }
    acc[3 * i] = ax; acc[3 * i + 1] = ay; acc[3 * i + 2] = az;
  }
  (void)nthreads;
  return 0;
}

ORACLE_API void oracle_kick_drift_f64(int n, double *pos, double *vel, const double *acc, double dt) {
  for (int i = 0; i < 3 * n; ++i) {
    vel[i] = vel[i] + dt * acc[i];
    pos[i] = pos[i] + dt * vel[i];
  }
}

/*
 * Total kinetic and potential energy in fp64: KE = 1/2 sum m v^2, PE = -G sum_{i<j} m_i m_j / sqrt(d^2+eps2).
 * Coincident pairs (d2 == 0) are skipped like the pair law skips them.
 */
ORACLE_API void oracle_energy_f64(int n, const double *pos, const double *vel, const double *mass,
                                  double g, double eps2, double *ke, double *pe, int nthreads) {
  double k = 0.0, u = 0.0;
#ifdef _OPENMP
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : k, u) num_threads(nthreads)
#endif
  for (int i = 0; i < n; ++i) {
    k += 0.5 * mass[i] * (vel[3 * i] * vel[3 * i] + vel[3 * i + 1] * vel[3 * i + 1] + vel[3 * i + 2] * vel[3 * i + 2]);
    for (int j = i + 1; j < n; ++j) {
      double dx = pos[3 * j] - pos[3 * i], dy = pos[3 * j + 1] - pos[3 * i + 1], dz = pos[3 * j + 2] - pos[3 * i + 2];
      double d2 = dx * dx + dy * dy + dz * dz + eps2;
      if (d2 == 0.0) continue;
      u -= g * mass[i] * mass[j] / sqrt(d2);
    }
  }
  (void)nthreads;
  *ke = k; *pe = u;
}

ORACLE_API int oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
