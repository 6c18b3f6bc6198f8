/* The child centre of Octree::Add (OctreeSearch.h:71-73) is float(double(o) +- double(Size) * 0.5) and the child's Size
 * float(0.5 * double(Size)) (.h:74).  The device's key computation (csrc/bh_common.h, descend_level) takes them as the plain
 * fp32 o +- 0.5f * Size and 0.5f * Size whenever Size >= 2^-100: this program checks that the two agree in every bit on
 * random operand pairs of all exponent distances (argv[1] = how many, default 2e7).  IEEE arithmetic only: what holds here
 * (SSE) holds on the GPU. */
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
static uint64_t s = 88172645463325252ull;
static uint64_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static float f_from(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }
int main(int argc, char **argv) {
  long bad = 0, n = 0;
  const long count = argc > 1 ? atol(argv[1]) : 20000000L;
  for (long it = 0; it < count; ++it) {
    uint32_t eb = (uint32_t)(rnd() % 254 + 1);             // exponent of size: normal
    if (eb < 27) continue;                                 // size >= 2^-100
    uint32_t sb = (eb << 23) | (uint32_t)(rnd() & 0x7FFFFF);
    volatile float size = f_from(sb);
    // o: random exponent anywhere (including denormal/zero), random sign; bias towards nearby exponents
    int mode = rnd() % 4;
    uint32_t oe = mode == 0 ? (uint32_t)(rnd() % 255) : (uint32_t)((int)eb + (int)(rnd() % 61) - 30);
    if ((int)oe < 0) oe = 0; if (oe > 254) oe = 254;
    uint32_t om = (uint32_t)(rnd() & 0x7FFFFF);
    if (rnd() % 16 == 0) om = 0; if (rnd() % 16 == 0) om = 0x7FFFFF;
    uint32_t ob = ((uint32_t)(rnd() & 1) << 31) | (oe << 23) | om;
    volatile float o = f_from(ob);
    for (int sg = 0; sg < 2; ++sg) {
      volatile float ref = (float)((double)o + (double)size * (sg ? 0.5 : -0.5));
      volatile float refs = (float)(0.5 * (double)size);
      volatile float hs = 0.5f * size;
      volatile float got = sg ? o + hs : o - hs;
      if (memcmp((void *)&ref, (void *)&got, 4) || memcmp((void *)&refs, (void *)&hs, 4)) { if (bad < 5) printf("bad o=%a size=%a ref=%a got=%a\n", o, size, ref, got); ++bad; }
      ++n;
    }
  }
  printf("checked %ld, bad %ld\n", n, bad);
  return bad != 0;
}
