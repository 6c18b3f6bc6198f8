// nbody_actor.hpp — plain-C++ mirror of the reference's UE4 actor AOctreeSearch
// (/root/reference/Source/NBody/OctreeSearch.h:111-149, OctreeSearch.cpp) on top of the C-ABI in
// nbody.h.  Same member names, defaults, call order and silent-guard error behaviour; UE4 types are
// replaced by standard ones (TArray -> std::vector, FVector -> float[3], DrawDebug* -> callbacks).
// A real UE4 adapter derives from AActor and forwards to this class.
//
// Header-only: needs only nbody.h and libnbody_amd.so.
#pragma once

#include <cstdint>
#include <functional>
#include <vector>

#include "nbody.h"

namespace nbody {

using FParticle = nbody_particle;   // OctreeSearch.h:8-18

class OctreeSearchActor {
 public:
  // ---- reference members (OctreeSearch.h:117-127) ----
  float Size = 0.0f;                    // .h:117
  std::vector<FParticle> Particles;     // .h:118  (host mirror of the device state, see MirrorParticles; empty when the
                                        //          records live in the host's own array: AllocateParticles)
  bool Initialized = false;             // .h:121
  bool ShowOctree = false;              // .h:124  BlueprintReadWrite
  float PhDeltaTime = 0.01f;            // .h:127  BlueprintReadWrite; default from the ctor, .cpp:8

  // ---- build-defined knobs (no reference counterpart) ----
  // Opening angle: the argument the reference hard-codes in `ComputeForces(&Particles[i], 1.0)` (.cpp:85), and the
  // default here too, so that an actor swapped in for AOctreeSearch reproduces its trajectories and its ShowOctree
  // boxes without further ado: the reference's own tree build, upsweep and walk run on the device (fp32, one device or a
  // device list; other precisions report NBODY_ERR_UNSUPPORTED in LastStatus).  Set Theta = 0 for the theta -> 0
  // limit of that walk — exact O(N^2) all-pairs, the hot path this engine exists for (any precision, any device list).
  float Theta = 1.0f;
  uint64_t Seed = 0x4E426F6479ull;      // CreateSpacePoints' generator seed (the reference is unseeded)
  float ActorLocation[3] = {0, 0, 0};   // GetActorLocation(), .cpp:64
  bool MirrorParticles = true;          // refresh `Particles` after every Tick, as the reference's TArray is live
  bool DrawInTick = true;          // Tick ends with DrawOctreeBoxes() (.cpp:33); a host that draws through its own DrawOctreeBoxes(Octree*) turns it off
  int Device = 0;
  std::vector<int32_t> Devices;         // non-empty: share the bodies over these GPUs (nbody_create_multi; any Theta: at Theta > 0 every
                                        // device builds the whole tree and walks its slice — the frames equal one device's in every byte)
  int Precision = NBODY_PREC_F32;
  double G = 1.0e4;                     // .h:104
  double Eps = 0.0;
  int LastStatus = NBODY_OK;            // last C-ABI return code (the reference's methods are void)

  // Where the records live.  Unset: in `Particles` above.  Set: the host's own array — a UE4 adapter hands out its
  // TArray<FParticle>'s storage (`Particles.SetNumUninitialized(n); return Particles.GetData();`), the device then writes every
  // frame's records straight into THAT memory (it is page-locked for the context: nbody_pin_host_buffer) and nothing is copied
  // on the host.  Called with the number of records whenever the actor needs storage for a new scene; the array must keep its
  // address until the next call (or CleanParticles).  A host that resizes or reallocates its array tells the actor with
  // SetParticles(data, n); one that only edits records with PushParticles().
  std::function<FParticle *(size_t n)> AllocateParticles;

  // Renderer hand-off (OctreeSearch.cpp:24,40-41): FlushPersistentDebugLines, DrawDebugPoint(Position, 10.0, Black),
  // DrawDebugBox(Origin, (Size,Size,Size), Red) per occupied leaf when ShowOctree (only with Theta > 0: at theta = 0
  // no tree exists).
  std::function<void()> OnFlushPersistentDebugLines;
  std::function<void(const float position[3], float point_size)> OnDrawDebugPoint;
  std::function<void(const float origin[3], float size)> OnDrawDebugBox;

  OctreeSearchActor() = default;        // .cpp:8
  OctreeSearchActor(const OctreeSearchActor &) = delete;
  OctreeSearchActor &operator=(const OctreeSearchActor &) = delete;
  ~OctreeSearchActor() { nbody_destroy(ctx_); }

  void BeginPlay() {}                   // .cpp:15-18

  // .cpp:58-72
  void CreateSpacePoints(int32_t N, float SizeArg = 200.0f) {
    if (N <= 0) { LastStatus = NBODY_ERR_INVALID; return; }
    Size = SizeArg;
    std::vector<float> posm(4 * (size_t)N), vel(4 * (size_t)N);
    LastStatus = nbody_ic_reference_box(N, SizeArg, ActorLocation, Seed, posm.data(), vel.data());
    if (LastStatus) return;
    Release();                          // before the records give up their (pinned) storage
    if (!Storage((size_t)N)) return;
    for (int32_t i = 0; i < N; ++i) {
      FParticle &p = data_[(size_t)i];
      p = FParticle{};
      p.Mass = posm[4 * (size_t)i + 3];
      for (int k = 0; k < 3; ++k) { p.Position[k] = posm[4 * (size_t)i + k]; p.Velocity[k] = vel[4 * (size_t)i + k]; }
    }
    Upload();
  }

  // Build-defined: explicit initial state instead of the random one (same post-conditions as CreateSpacePoints).
  // With AllocateParticles set and `p` the host's own array (same address as the hook returns for N records) nothing is
  // copied: the array IS the record storage from here on.
  void SetParticles(const FParticle *p, int32_t N) {
    if (!p || N <= 0) { LastStatus = NBODY_ERR_INVALID; return; }
    std::vector<FParticle> copy(p, p + N);   // `p` may point into the storage that is about to be released / reallocated
    Release();
    if (!Storage((size_t)N)) return;
    if (data_ != p) std::copy(copy.begin(), copy.end(), data_);
    Upload();
  }

  // The host has edited records in place (the reference's `Particles` IS the state, .h:118: whatever code changes
  // Particles[i] between two Ticks changes the simulation, .cpp:28-31).  Here the state lives on the device and the array is
  // its mirror: an edit is overwritten by the next frame unless it is pushed.  Keeps the history (step count, the next
  // tree's root centre = the previous tree's CoM, .cpp:77-79): nbody_push_particles.
  void PushParticles() {
    if (!Initialized) return;
    LastStatus = nbody_push_particles(ctx_, data_, sizeof(FParticle), (int32_t)num_);
    if (LastStatus == NBODY_OK) dirty_ = false;
  }

  // The host is about to resize or reallocate the array behind AllocateParticles (TArray::Add / SetNum / Empty + refill ...): the
  // context lets go of the storage FIRST — a page-locked registration must not outlive its allocation (hipHostUnregister on memory
  // the allocator has already handed back is undefined).  Afterwards the host tells the actor about the new storage with
  // SetParticles(data, n) before anything else (a UE4 adapter does both in its mutators / at the top of its Tick).
  void ReleaseStorage() {
    if (ctx_ && data_ && pinned_) (void)nbody_unpin_host_buffer(ctx_, data_);
    pinned_ = false;
  }
  bool StoragePinned() const { return pinned_; }
  // ... and the other way round: the storage is where it was after all (same address, same size): page-lock it again
  void PinStorage() {
    if (ctx_ && data_ && !pinned_) pinned_ = nbody_pin_host_buffer(ctx_, data_, num_ * sizeof(FParticle)) == NBODY_OK;
  }

  // The records (`Particles`, or the host's array behind AllocateParticles) and their number
  FParticle *ParticleData() { return data_; }
  const FParticle *ParticleData() const { return data_; }
  size_t NumParticles() const { return num_; }

  // .cpp:47-56
  void ComputeCubeSize() {
    if (!Initialized) return;
    float s = 0.0f;
    LastStatus = nbody_get_bounds(ctx_, &s);
    if (LastStatus == NBODY_OK) Size = s;
  }

  // .cpp:74-89 — the force pass (tree build + walk in the reference; all-pairs kernel here).
  void CreateOctree() {
    if (!Initialized) return;
    LastStatus = nbody_set_theta(ctx_, Theta);
    if (LastStatus) return;
    LastStatus = nbody_compute_forces(ctx_);
    forces_fresh_ = LastStatus == NBODY_OK;
  }

  // .cpp:21-34
  void Tick(float /*DeltaSeconds: ignored by the reference too*/) {
    if (OnFlushPersistentDebugLines) OnFlushPersistentDebugLines();            // .cpp:24
    if (PhDeltaTime > 0 && Initialized) {                                      // .cpp:25 (+ guards .cpp:49,76)
      LastStatus = nbody_set_theta(ctx_, Theta);
      if (LastStatus == NBODY_OK && MirrorParticles) {
        // .cpp:26-31 and the mirror DrawOctreeBoxes reads, in one call with one host synchronisation
        float s = Size;
        LastStatus = nbody_tick(ctx_, PhDeltaTime, &s, data_, sizeof(FParticle));
        if (LastStatus == NBODY_OK) { Size = s; dirty_ = false; forces_fresh_ = true; }
      } else if (LastStatus == NBODY_OK) {
        ComputeCubeSize();                                                     // .cpp:26
        LastStatus = nbody_step(ctx_, PhDeltaTime, 1);                         // .cpp:27-31
        if (LastStatus == NBODY_OK) { dirty_ = true; forces_fresh_ = true; }
      }
    }
    if (DrawInTick) DrawOctreeBoxes();                                         // .cpp:33
  }

  // Has a force pass produced something DrawOctreeBoxes can draw (the reference: ParticleOctree != NULL, .cpp:38)?
  bool HasTree() const { return Initialized && forces_fresh_; }

  // .cpp:36-45 — one DrawDebugPoint per body.
  void DrawOctreeBoxes() {
    if (!Initialized || !forces_fresh_) return;   // the reference draws from the tree: nothing before the first force pass (.cpp:38)
    if (MirrorParticles) SyncParticles(); else SyncPositions();
    if (Theta > 0.0f) {
      // the reference walks its tree depth first, children 0..7, and at every occupied leaf draws the box (if
      // ShowOctree) and then the point (.cpp:39-41): same order, same interleaving
      const bool boxes = ShowOctree && OnDrawDebugBox;
      if (!boxes && !OnDrawDebugPoint) return;
      order_.resize(num_);
      if (nbody_bh_leaf_order(ctx_, order_.data()) == NBODY_OK) {
        if (boxes) {
          boxes_.resize(4 * num_);
          if (nbody_bh_leaf_boxes(ctx_, boxes_.data(), 16) != NBODY_OK) boxes_.clear();
        }
        for (int32_t i : order_) {
          if (boxes && !boxes_.empty()) OnDrawDebugBox(&boxes_[4 * (size_t)i], boxes_[4 * (size_t)i + 3]);          // .cpp:40
          if (OnDrawDebugPoint) OnDrawDebugPoint(data_[(size_t)i].Position, 10.0f);                                    // .cpp:41
        }
        return;
      }
    }
    // theta = 0: there is no tree; bodies in index order, no boxes
    if (OnDrawDebugPoint)
      for (size_t i = 0; i < num_; ++i) OnDrawDebugPoint(data_[i].Position, 10.0f);
  }

  // .cpp:91-97
  void CleanParticles() {
    Initialized = false;
    nbody_destroy(ctx_);
    ctx_ = nullptr;
    forces_fresh_ = false;
    dirty_ = false;
    Particles.clear();
    data_ = nullptr; num_ = 0;          // (a host array behind AllocateParticles is the host's to empty)
    pinned_ = false;
  }

  // Pull the whole device state into the records.
  void SyncParticles() {
    if (!ctx_ || !dirty_) return;
    LastStatus = nbody_get_particles(ctx_, data_, sizeof(FParticle));
    if (LastStatus == NBODY_OK) dirty_ = false;
  }

  nbody_ctx *Context() const { return ctx_; }

 private:
  void SyncPositions() {
    if (!ctx_ || !dirty_) return;
    LastStatus = nbody_get_positions(ctx_, data_->Position, sizeof(FParticle), 0, (int32_t)num_);
  }

  // storage for n records: the host's array if it hands one out, `Particles` otherwise
  bool Storage(size_t n) {
    if (AllocateParticles) {
      Particles.clear();
      data_ = AllocateParticles(n);
      if (!data_) { num_ = 0; LastStatus = NBODY_ERR_NOMEM; return false; }
    } else {
      Particles.assign(n, FParticle{});
      data_ = Particles.data();
    }
    num_ = n;
    return true;
  }

  void Release() {
    nbody_destroy(ctx_);                // also unpins `Particles`
    ctx_ = nullptr;
    pinned_ = false;
    Initialized = false;
  }

  void Upload() {
    Release();
    nbody_params p;
    nbody_default_params(&p);
    p.n_total = (int32_t)num_;
    p.device = Device;
    p.precision = Precision;
    p.G = G;
    p.eps = Eps;
    LastStatus = Devices.empty() ? nbody_create(&p, &ctx_)
                                 : nbody_create_multi(&p, Devices.data(), (int32_t)Devices.size(), &ctx_);
    if (LastStatus) return;
    LastStatus = nbody_set_particles(ctx_, data_, sizeof(FParticle), (int32_t)num_);
    if (LastStatus) return;
    // the per-frame records land in the storage straight from the device (an optimisation only: unpinned memory works the
    // same).  The storage must keep its address while the actor is initialised, as the reference's TArray does between
    // CreateSpacePoints and CleanParticles.
    pinned_ = nbody_pin_host_buffer(ctx_, data_, num_ * sizeof(FParticle)) == NBODY_OK;
    Initialized = true;                                                        // .cpp:71
    forces_fresh_ = false;
    dirty_ = false;
  }

  nbody_ctx *ctx_ = nullptr;
  FParticle *data_ = nullptr;           // the records: Particles.data() or the host's array
  size_t num_ = 0;
  std::vector<float> boxes_;
  std::vector<int32_t> order_;
  bool forces_fresh_ = false;
  bool dirty_ = false;
  bool pinned_ = false;                 // the records' storage is page-locked for ctx_ (nbody_pin_host_buffer)
};

}  // namespace nbody
