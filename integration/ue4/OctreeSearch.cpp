// integration/ue4/OctreeSearch.cpp — see OctreeSearch.h next to it.  Every method keeps the reference's observable
// behaviour (silent guards, what gets drawn and in which order); its body is a call into nbody::OctreeSearchActor.
// NOT COMPILED IN THIS REPOSITORY (no Unreal Engine 4.9 here).
#include "NBody.h"
#include "OctreeSearch.h"
#include "DrawDebugHelpers.h"

#include <cstring>

AOctreeSearch::AOctreeSearch() : Size(0), ParticleOctree(NULL), Initialized(false), ShowOctree(false), PhDeltaTime(0.01), Theta(1.0f)
{
  Engine.DrawInTick = false;                                  // Tick draws through DrawOctreeBoxes(ParticleOctree), as the reference does
  // The records live in THIS actor's TArray (OctreeSearch.h:118): the engine asks for storage here, page-locks it for its
  // context and lets the device write every frame's FParticle records straight into it.
  Engine.AllocateParticles = [this](size_t N) {
    Particles.SetNumUninitialized((int32)N);
    return reinterpret_cast<nbody::FParticle*>(Particles.GetData());
  };
  PrimaryActorTick.bCanEverTick = true;                       // OctreeSearch.cpp:11
}

void AOctreeSearch::BeginPlay()
{
  Super::BeginPlay();
  // What the reference draws itself (OctreeSearch.cpp:24, 40, 41) arrives through callbacks, in the same order: per
  // occupied leaf, depth first, the box (if ShowOctree) and then the point.
  Engine.OnFlushPersistentDebugLines = [this]() { FlushPersistentDebugLines(GetWorld()); };
  Engine.OnDrawDebugBox = [this](const float* Origin, float HalfWidth) {
    DrawDebugBox(GetWorld(), FVector(Origin[0], Origin[1], Origin[2]), FVector(HalfWidth, HalfWidth, HalfWidth), FColor::Red, true);
  };
  Engine.OnDrawDebugPoint = [this](const float* Position, float PointSize) {
    DrawDebugPoint(GetWorld(), FVector(Position[0], Position[1], Position[2]), PointSize, FColor::Black, true);
  };
}

void AOctreeSearch::PushKnobs()
{
  Engine.PhDeltaTime = PhDeltaTime;
  Engine.ShowOctree = ShowOctree;
  Engine.Theta = Theta;
  Engine.Devices.assign(Devices.GetData(), Devices.GetData() + Devices.Num());
}

void AOctreeSearch::PullMirror()
{
  Size = Engine.Size;
  Initialized = Engine.Initialized;
  // the reference's pointer is non-NULL from the first CreateOctree until CleanParticles (OctreeSearch.cpp:79, 94-95)
  ParticleOctree = Engine.HasTree() ? reinterpret_cast<Octree*>(&Engine) : NULL;
  // nothing to copy: the frame's records are in Particles already (Engine.AllocateParticles)
}

// The host resized or reallocated `Particles` (Add / SetNum / Empty + refill ...): what is in the array now is the scene.
void AOctreeSearch::AdoptStorage()
{
  if (!Engine.Initialized) return;
  const nbody::FParticle* Data = reinterpret_cast<const nbody::FParticle*>(Particles.GetData());
  if (Data == Engine.ParticleData() && (size_t)Particles.Num() == Engine.NumParticles()) {
    Engine.PinStorage();                                      // (released, and the array stayed where it was after all)
    return;
  }
  // (Here the old storage is gone already.  Had the caller not announced the change with ReleaseStorage(), the old range would
  //  still be page-locked for the context that SetParticles is about to destroy: unregistering memory the allocator has handed
  //  back is undefined — hence the rule in OctreeSearch.h.)
  if (Particles.Num() == 0) { Engine.CleanParticles(); return; }
  Engine.SetParticles(Data, Particles.Num());                 // re-creates the context on (and re-pins) the new storage
}

void AOctreeSearch::ReleaseStorage()
{
  Engine.ReleaseStorage();
}

void AOctreeSearch::PushParticles()
{
  AdoptStorage();
  Engine.PushParticles();
}

// OctreeSearch.cpp:21-34: flush, (if PhDeltaTime > 0) bounds + force pass + kick-drift, draw.
void AOctreeSearch::Tick(float DeltaSeconds)
{
  Super::Tick(DeltaSeconds);
  PushKnobs();
  AdoptStorage();
  Engine.Tick(DeltaSeconds);                                  // DeltaSeconds is ignored, as in the reference
  PullMirror();
  DrawOctreeBoxes(ParticleOctree);                            // OctreeSearch.cpp:33 (runs when paused too)
}

// OctreeSearch.cpp:36-45: per occupied leaf, depth first with children 0..7, the box (if ShowOctree) and the point — drawn
// by the callbacks BeginPlay installed, from what the device's tree says (nbody_bh_leaf_order / nbody_bh_leaf_boxes).
void AOctreeSearch::DrawOctreeBoxes(Octree* Oct)
{
  if (Oct == NULL) return;                                    // OctreeSearch.cpp:38
  Engine.ShowOctree = ShowOctree;
  Engine.DrawOctreeBoxes();
}

// OctreeSearch.cpp:47-56
void AOctreeSearch::ComputeCubeSize()
{
  Engine.ComputeCubeSize();
  Size = Engine.Size;
}

// OctreeSearch.cpp:58-72.  The reference draws from the engine's global unseeded RNG; the distribution is the same here,
// the generator is seeded (Engine.Seed) so that a scene can be replayed.
void AOctreeSearch::CreateSpacePoints(int32 N, float SizeArg)
{
  const FVector L = GetActorLocation();
  Engine.ActorLocation[0] = L.X; Engine.ActorLocation[1] = L.Y; Engine.ActorLocation[2] = L.Z;
  PushKnobs();
  Engine.CreateSpacePoints(N, SizeArg);
  PullMirror();
}

// OctreeSearch.cpp:74-89: the force pass alone (the reference's Blueprints never call it; Tick does).
void AOctreeSearch::CreateOctree()
{
  PushKnobs();
  Engine.CreateOctree();
  Engine.SyncParticles();
  PullMirror();
}

// OctreeSearch.cpp:91-97
void AOctreeSearch::CleanParticles()
{
  Engine.CleanParticles();                                    // destroys the context (and unpins the array) first ...
  Particles.Empty();                                          // ... then OctreeSearch.cpp:96
  PullMirror();
}
