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
  std::printf(g_fail ? "actor parity: %d FAILED\n" : "actor parity: ok\n", g_fail);
  return g_fail ? 1 : 0;
}
