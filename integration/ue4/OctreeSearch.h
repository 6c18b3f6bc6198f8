// integration/ue4/OctreeSearch.h — the UE4 side of the drop-in: AOctreeSearch with the public surface the Blueprints of
// Milias/ParallelNbody bind to (BP_NBodyHUD spawns the actor and calls CreateSpacePoints / SetActorTickEnabled;
// BP_ScreenUI calls CleanParticles / CreateSpacePoints and writes PhDeltaTime / ShowOctree), and the physics handed to
// libnbody_amd.so through include/nbody_actor.hpp.  It replaces Source/NBody/OctreeSearch.h of the reference module
// (reference lines cited as OctreeSearch.h:N / OctreeSearch.cpp:N); the Octree class and the per-particle loops of the
// reference are gone — their work happens on the GPU.
//
// NOT COMPILED IN THIS REPOSITORY: Unreal Engine 4.9 and UnrealBuildTool are not available here.  What the file
// forwards to is compiled and tested (tests/cpp/actor_parity.cpp, tests/test_actor_gpu.py).
// Module wiring, Source/NBody/NBody.Build.cs:
//     PublicIncludePaths.Add(Path.Combine(ThirdPartyPath, "nbody_amd", "include"));
//     PublicAdditionalLibraries.Add(Path.Combine(ThirdPartyPath, "nbody_amd", "lib", "libnbody_amd.so"));
#pragma once

#include "GameFramework/Actor.h"
#include "nbody_actor.hpp"            // nbody::OctreeSearchActor, nbody::FParticle (same 40-byte layout as the USTRUCT below)
#include "OctreeSearch.generated.h"

// OctreeSearch.h:8-18.  Kept as a USTRUCT so that existing Blueprint references to the type stay valid.
USTRUCT()
struct FParticle {
  GENERATED_USTRUCT_BODY()

  float Mass;
  FVector Position;
  FVector Velocity;
  FVector Acceleration;

  FParticle() : Mass(0), Position(FVector::ZeroVector), Velocity(FVector::ZeroVector), Acceleration(FVector::ZeroVector) {}
};
static_assert(sizeof(FParticle) == sizeof(nbody::FParticle), "FParticle must stay 40 bytes: Mass, Position, Velocity, Acceleration");

class Octree;                                // OctreeSearch.h:21-109 — opaque here: see the header comment

UCLASS()
class NBODY_API AOctreeSearch : public AActor
{
  GENERATED_BODY()

public:
  // ---- the reference's members (OctreeSearch.h:117-127) ----
  float Size;                               // half-width of the scene, refreshed by every Tick (ComputeCubeSize)
  TArray<FParticle> Particles;              // OctreeSearch.h:118.  The array's own storage is where the device writes every frame's
                                            // records (it is page-locked for the context; nothing is copied on the host).  The
                                            // simulation state lives on the device: code that EDITS Particles[i] calls
                                            // PushParticles() afterwards; code that RESIZES or reallocates the array (Add /
                                            // SetNum / Empty + refill) calls ReleaseStorage() first — a page-locked range must
                                            // not outlive its allocation — and the next Tick (or PushParticles) sees the new
                                            // storage and re-creates the engine on it.
  Octree* ParticleOctree;                   // OctreeSearch.h:119: NULL until the first force pass and after CleanParticles
                                            // (OctreeSearch.cpp:8, 95); otherwise a token for the device's current tree
  bool Initialized;

  UPROPERTY(BlueprintReadWrite)
  bool ShowOctree;

  UPROPERTY(BlueprintReadWrite)
  float PhDeltaTime;

  // ---- new, optional (defaults reproduce the reference) ----
  // Opening angle of the tree walk; the reference hard-codes 1.0 (OctreeSearch.cpp:85).  0 = exact O(N^2) all-pairs.
  UPROPERTY(BlueprintReadWrite)
  float Theta;

  // GPUs to share the bodies over (empty = device 0; nbody_create_multi).  Any Theta: at Theta > 0 every device builds the whole
  // tree from its copy of the positions and walks its own slice — the frames equal one device's in every byte.
  TArray<int32> Devices;

  AOctreeSearch();

  virtual void BeginPlay() override;
  virtual void Tick(float DeltaSeconds) override;

  void DrawOctreeBoxes(Octree* Oct);        // OctreeSearch.h:138, OctreeSearch.cpp:36-45: Oct == NULL draws nothing
  void ComputeCubeSize();

  UFUNCTION(BlueprintCallable, Category = "Octree")
  void CreateSpacePoints(int32 N, float Size = 200);

  UFUNCTION(BlueprintCallable, Category = "Octree")
  void CreateOctree();

  UFUNCTION(BlueprintCallable, Category = "Octree")
  void CleanParticles();

  // new: records of `Particles` edited by the host since the last Tick -> the device (in the reference the edit alone would
  // do: the Tick integrates the array in place, OctreeSearch.cpp:28-31).  Keeps the history: the next tree is still rooted at the
  // previous tree's centre of mass (OctreeSearch.cpp:77-79).
  UFUNCTION(BlueprintCallable, Category = "Octree")
  void PushParticles();

  // new: call BEFORE code resizes or reallocates `Particles`: the engine unpins the array's storage while it is still allocated.
  // (CreateSpacePoints and CleanParticles do it themselves.)  Frames keep arriving; the new storage is adopted by the next Tick.
  UFUNCTION(BlueprintCallable, Category = "Octree")
  void ReleaseStorage();

private:
  nbody::OctreeSearchActor Engine;          // owns the nbody_ctx; never touched from Blueprints
  void PullMirror();                        // Engine's Size / Initialized / tree token -> the members above (no record is copied)
  void PushKnobs();                         // PhDeltaTime / ShowOctree / Theta / Devices -> Engine
  void AdoptStorage();                      // the TArray was resized or reallocated by the host: the engine moves to the new storage
};
