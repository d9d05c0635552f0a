// TEST INFRASTRUCTURE ONLY (oracle/): known-answer generator for the reference's
// in-tree transforms.  This harness is OUR code; it #includes the reference's template
// "library" (trunk/src/dwt2d.cpp, 5_3.cpp, Haar.cpp) by path at build time -- the reference
// sources are never copied into this repository.  Built into oracle/_ref/dwt_probe by
// oracle/Makefile when /root/reference is present.
//
// usage: dwt_probe <53|haar> <a|s> H W levels  < plane.int16le  > plane.int16le
//   a = dwt2d::analyze (dwt2d.cpp:76-119), s = dwt2d::synthesize (dwt2d.cpp:128-175)
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "5_3.cpp"
#include "Haar.cpp"
#include "dwt2d.cpp"

template <class D> static void run(D &d, char mode, short **rows, int H, int W, int levels) {
  d.set_max_line_size(16384);
  if (mode == 'a') d.analyze(rows, H, W, levels);
  else d.synthesize(rows, H, W, levels);
}

int main(int argc, char **argv) {
  if (argc != 6) { fprintf(stderr, "usage: %s <53|haar> <a|s> H W levels\n", argv[0]); return 2; }
  int H = atoi(argv[3]), W = atoi(argv[4]), levels = atoi(argv[5]);
  short *buf = (short *)malloc(sizeof(short) * H * W);
  short **rows = (short **)malloc(sizeof(short *) * H);
  if (fread(buf, sizeof(short), (size_t)H * W, stdin) != (size_t)H * W) { fprintf(stderr, "short read\n"); return 1; }
  for (int y = 0; y < H; y++) rows[y] = buf + (size_t)y * W;
  if (!strcmp(argv[1], "53")) { dwt2d<short, _5_3<short> > d; run(d, argv[2][0], rows, H, W, levels); }
  else { dwt2d<short, Haar<short> > d; run(d, argv[2][0], rows, H, W, levels); }
  fwrite(buf, sizeof(short), (size_t)H * W, stdout);
  return 0;
}
