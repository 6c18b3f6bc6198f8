/* The plain-C host of INTEGRATION.md section 3, as a program: nbody_actor.h driven the way BP_NBodyHUD drives AOctreeSearch
 * (spawn, CreateSpacePoints(2000, 1000), Tick every frame), at the shipped opening angle and at the exact limit.
 * Exit code 0 = every frame drew every body and nothing failed; 2 = no HIP device. */
#include <stdio.h>

#include "nbody.h"
#include "nbody_actor.h"

static long points, flushes;
static void draw_point(void *user, const float pos[3], float size) { (void)user; (void)size; if (pos[0] == pos[0]) ++points; }
static void flush(void *user) { (void)user; ++flushes; }

int main(void) {
  if (nbody_device_count() <= 0) { printf("actor demo: no HIP device\n"); return 2; }
  for (int pass = 0; pass < 2; ++pass) {
    const float theta = pass == 0 ? 1.0f : 0.0f;               /* the shipped opening angle; 0 = exact all-pairs */
    nbody_actor *a = nbody_actor_create();                     /* AOctreeSearch() */
    nbody_actor_set_draw_callbacks(a, flush, draw_point, NULL);
    nbody_actor_create_space_points(a, 2000, 1000.0f);         /* what BP_NBodyHUD does at BeginPlay */
    nbody_actor_set_theta(a, theta);
    points = flushes = 0;
    for (int frame = 0; frame < 60; ++frame) nbody_actor_tick(a, 1.0f / 60);
    const int failed = nbody_actor_last_status(a) != 0 || points != 60L * 2000 || flushes != 60 ||
                       !(nbody_actor_get_size(a) > 0.0f);
    printf("actor demo: theta %.1f  %ld points in %ld frames  Size %.3f  status %d\n", theta, points, flushes,
           nbody_actor_get_size(a), nbody_actor_last_status(a));
    nbody_actor_clean_particles(a);
    nbody_actor_destroy(a);
    if (failed) return 1;
  }
  printf("actor demo: ok\n");
  return 0;
}
