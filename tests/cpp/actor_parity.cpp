// Parity test in the reference's own language: drives nbody::OctreeSearchActor (include/nbody_actor.hpp) the way the
// reference's Blueprints drive AOctreeSearch — CreateSpacePoints / Tick / ShowOctree / PhDeltaTime / CleanParticles —
// and checks every frame against the CPU oracle's Tick (oracle/nbody_oracle.c).  Built with g++ against
// libnbody_amd.so and libnbody_oracle.so by tests/test_cpp_actor_gpu.py; needs a GPU to run.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "nbody_actor.hpp"

struct oracle_particle { float Mass, Position[3], Velocity[3], Acceleration[3]; };
extern "C" int oracle_tick_aos_f32(int n, oracle_particle *p, float dt, float theta, double g, int pow_mode,
                                   float root_com[3], float *size_io);

static int g_fail = 0;
#define CHECK(cond, ...) do { if (!(cond)) { std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); ++g_fail; } } while (0)

static double max_rel_acc(const std::vector<nbody::FParticle> &a, const std::vector<oracle_particle> &b, int *bit_equal) {
  double worst = 0.0;
  int same = 0;
  for (size_t i = 0; i < a.size(); ++i) {
    double num = 0, den = 0;
    for (int k = 0; k < 3; ++k) {
      const double d = (double)a[i].Acceleration[k] - (double)b[i].Acceleration[k];
      num += d * d; den += (double)b[i].Acceleration[k] * (double)b[i].Acceleration[k];
    }
    if (den > 0) worst = std::fmax(worst, std::sqrt(num / den));
    same += std::memcmp(a[i].Acceleration, b[i].Acceleration, 12) == 0;
  }
  if (bit_equal) *bit_equal = same;
  return worst;
}

int main() {
  static_assert(sizeof(nbody::FParticle) == 40 && sizeof(oracle_particle) == 40, "FParticle layout (OctreeSearch.h:8-18)");
  if (nbody_device_count() < 1) { std::printf("no HIP device\n"); return 2; }

  nbody::OctreeSearchActor actor;                       // AOctreeSearch(), .cpp:8
  CHECK(actor.Size == 0 && !actor.Initialized && !actor.ShowOctree && actor.PhDeltaTime == 0.01f, "ctor defaults");
  actor.Tick(0.016f);                                   // before CreateSpacePoints: silent no-op
  int points = 0, flushes = 0, boxes = 0;
  actor.OnFlushPersistentDebugLines = [&] { ++flushes; };
  actor.OnDrawDebugPoint = [&](const float *, float size) { points += size == 10.0f; };
  actor.OnDrawDebugBox = [&](const float *, float) { ++boxes; };

  CHECK(actor.Theta == 1.0f, "the opening angle defaults to the reference's hard-coded 1.0 (.cpp:85)");
  actor.Seed = 1;
  actor.Theta = 0.0f;                                   // first the exact all-pairs limit
  actor.CreateSpacePoints(2000, 1000.0f);               // BP_NBodyHUD BeginPlay
  CHECK(actor.Initialized && actor.LastStatus == NBODY_OK && actor.Particles.size() == 2000, "CreateSpacePoints status %d", actor.LastStatus);
  CHECK(actor.Particles[0].Mass == 5000.0f && actor.Particles[0].Position[0] == 0.0f, "body 0 pinned (.cpp:68-70)");

  // ---- theta = 0: the all-pairs hot path, frame by frame against the oracle's index-order Tick ----
  std::vector<oracle_particle> ref(2000);
  std::memcpy(ref.data(), actor.Particles.data(), 2000 * 40);
  float com[3] = {0, 0, 0}, size = 0.0f;
  for (int frame = 0; frame < 3; ++frame) {
    actor.Tick(0.016f);
    oracle_tick_aos_f32(2000, ref.data(), 0.01f, -1.0f, 1.0e4, 0, com, &size);
    const double e = max_rel_acc(actor.Particles, ref, nullptr);
    CHECK(actor.LastStatus == NBODY_OK && e < (frame == 0 ? 2e-5 : 1e-3), "theta=0 frame %d: rel acc err %.3e", frame, e);
    CHECK(std::fabs(actor.Size - size) <= 1e-6f * size, "ComputeCubeSize %g vs %g", actor.Size, size);   // positions agree to ~1e-9 after a frame
    if (frame == 0) std::printf("theta = 0   frame 0: max rel acceleration error vs oracle %.3e\n", e);
  }
  CHECK(flushes == 3 && points == 3 * 2000 && boxes == 0, "draw calls: flush %d points %d boxes %d", flushes, points, boxes);

  // ---- pause (BP_ScreenUI): PhDeltaTime = 0 freezes the physics but still draws (.cpp:25,33) ----
  const std::vector<nbody::FParticle> frozen = actor.Particles;
  actor.PhDeltaTime = 0.0f;
  actor.Tick(0.016f);
  CHECK(std::memcmp(frozen.data(), actor.Particles.data(), 2000 * 40) == 0 && points == 4 * 2000, "pause");
  actor.PhDeltaTime = 0.01f;

  // ---- theta = 1.0: the reference's shipped opening angle, its own tree on the device ----
  actor.CleanParticles();                               // Button_98: CleanParticles -> CreateSpacePoints
  CHECK(!actor.Initialized && actor.Particles.empty(), "CleanParticles");
  actor.Theta = 1.0f;
  actor.ShowOctree = true;
  actor.CreateSpacePoints(2000, 1000.0f);
  std::memcpy(ref.data(), actor.Particles.data(), 2000 * 40);
  com[0] = com[1] = com[2] = 0.0f; size = 0.0f;
  boxes = 0;
  for (int frame = 0; frame < 3; ++frame) {
    actor.Tick(0.016f);
    oracle_tick_aos_f32(2000, ref.data(), 0.01f, 1.0f, 1.0e4, 0, com, &size);
    int same = 0;
    const double e = max_rel_acc(actor.Particles, ref, &same);
    CHECK(actor.LastStatus == NBODY_OK && e < 1e-6 && same >= 1998, "theta=1 frame %d: rel err %.3e, %d/2000 bit-equal", frame, e, same);
    if (frame == 0) std::printf("theta = 1.0 frame 0: %d/2000 accelerations bit-equal to the oracle's tree walk (max rel %.1e)\n", same, e);
  }
  CHECK(boxes == 3 * 2000, "ShowOctree boxes %d", boxes);

  // ---- the host edits a body between two Ticks (the reference's Particles IS the state, OctreeSearch.h:118, .cpp:28-31) ----
  // without a push the device state wins: the next frame overwrites the edit ...
  const nbody::FParticle before = actor.Particles[7];
  actor.Particles[7].Velocity[0] += 1000.0f;
  actor.Tick(0.016f);
  oracle_tick_aos_f32(2000, ref.data(), 0.01f, 1.0f, 1.0e4, 0, com, &size);
  CHECK(std::fabs(actor.Particles[7].Velocity[0] - ref[7].Velocity[0]) < 1.0f && before.Mass == actor.Particles[7].Mass,
        "an edit that is not pushed does not reach the simulation");
  // ... with PushParticles it is the simulation's state from the next Tick on, history kept: the oracle's Tick on the same
  // edited records, with the SAME root centre (the previous tree's CoM, .cpp:77-79) and Size carried over
  for (std::vector<oracle_particle> *dst : {&ref}) {
    (*dst)[7].Velocity[0] += 1000.0f; (*dst)[7].Position[2] -= 25.0f; (*dst)[7].Mass = 4000.0f;
    (*dst)[1999].Position[0] = 3.0f * size;                       // a body moved outside the old bounds: Size must follow
  }
  actor.Particles[7].Velocity[0] += 1000.0f; actor.Particles[7].Position[2] -= 25.0f; actor.Particles[7].Mass = 4000.0f;
  actor.Particles[1999].Position[0] = ref[1999].Position[0];
  actor.PushParticles();
  CHECK(actor.LastStatus == NBODY_OK, "PushParticles status %d", actor.LastStatus);
  for (int frame = 0; frame < 2; ++frame) {
    actor.Tick(0.016f);
    oracle_tick_aos_f32(2000, ref.data(), 0.01f, 1.0f, 1.0e4, 0, com, &size);
    int same = 0;
    const double e = max_rel_acc(actor.Particles, ref, &same);
    CHECK(actor.LastStatus == NBODY_OK && e < 1e-6 && same >= 1998, "after PushParticles, frame %d: rel err %.3e, %d/2000 bit-equal", frame, e, same);
    CHECK(actor.Size == size, "Size after the edit %g vs %g", actor.Size, size);
    CHECK(actor.Particles[7].Mass == 4000.0f && std::fabs(actor.Particles[7].Position[2] - ref[7].Position[2]) <= 1e-4f * std::fabs(ref[7].Position[2]) + 1e-4f &&
          std::fabs(actor.Particles[7].Velocity[0] - ref[7].Velocity[0]) <= 1e-4f * std::fabs(ref[7].Velocity[0]), "the edited body");
    if (frame == 0) std::printf("PushParticles: edited bodies simulated on, %d/2000 accelerations bit-equal to the oracle's Tick of the edited state\n", same);
  }
  int64_t steps = 0;
  CHECK(nbody_steps_done(actor.Context(), &steps) == NBODY_OK && steps == 6, "history kept across the push: %lld steps", (long long)steps);

  // ---- the records in the HOST's own array (what the UE4 adapter does with its TArray<FParticle>): no host copy ----
  {
    std::vector<nbody::FParticle> mine;                 // stands in for AOctreeSearch::Particles
    int allocations = 0;
    nbody::OctreeSearchActor hosted, plain;
    hosted.AllocateParticles = [&](size_t n) { ++allocations; mine.resize(n); return mine.data(); };
    for (nbody::OctreeSearchActor *a : {&hosted, &plain}) { a->Seed = 3; a->CreateSpacePoints(2000, 1000.0f); }
    CHECK(hosted.Initialized && hosted.Particles.empty() && hosted.ParticleData() == mine.data() && mine.size() == 2000 && allocations == 1,
          "AllocateParticles: the records live in the host's array");
    for (int frame = 0; frame < 3; ++frame) { hosted.Tick(0.016f); plain.Tick(0.016f); }
    CHECK(hosted.LastStatus == NBODY_OK && std::memcmp(mine.data(), plain.Particles.data(), 2000 * 40) == 0 && hosted.Size == plain.Size,
          "frames delivered into the host's array equal the plain actor's");
    // the host grows its array (TArray::Add reallocates): the adapter hands the new storage over with SetParticles
    std::vector<nbody::FParticle> grown(mine);
    grown.push_back(grown[5]); grown.back().Position[0] += 17.0f;
    mine.clear(); mine.shrink_to_fit();
    hosted.SetParticles(grown.data(), (int32_t)grown.size());
    plain.SetParticles(grown.data(), (int32_t)grown.size());
    hosted.Tick(0.016f); plain.Tick(0.016f);
    CHECK(allocations == 2 && mine.size() == 2001 && hosted.NumParticles() == 2001 &&
          std::memcmp(mine.data(), plain.Particles.data(), 2001 * 40) == 0, "a resized host array is adopted");
    hosted.CleanParticles();
    CHECK(!hosted.Initialized && hosted.NumParticles() == 0 && mine.size() == 2001, "CleanParticles leaves the host's array to the host");
  }
  std::printf(g_fail ? "actor parity: %d FAILED\n" : "actor parity: ok\n", g_fail);
  return g_fail ? 1 : 0;
}
