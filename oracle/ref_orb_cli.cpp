// ref_orb_cli.cpp — TEST INFRASTRUCTURE: one extraction by the reference's own cslam::ORBextractor in a process whose global operator
// new is the monotonic one of ref_bump_new.cpp (a replacement operator new must be defined by the PROGRAM to apply to every module,
// libstdc++ included; inside a library loaded into python it cannot).  With addresses increasing in allocation order, the reference's
// pointer tie-break in DistributeOctTree (ORBextractor.cpp:852) is "node creation order", the tie rule the oracle and the product define.
// usage: orb_ref_cli <in.raw> <w> <h> <nfeatures> <scale> <nlevels> <iniTh> <minTh> <out.bin>
//   out.bin: int32 n, then n x {float x, y, size, angle, response; int32 octave}, then n x 32 descriptor bytes
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cslam/ORBextractor.h>

int main(int argc, char** argv) {
  if (argc != 10) { std::fprintf(stderr, "usage: %s in.raw w h nfeatures scale nlevels iniTh minTh out.bin\n", argv[0]); return 2; }
  const int w = std::atoi(argv[2]), h = std::atoi(argv[3]);
  std::vector<unsigned char> img((size_t)w * h);
  FILE* f = std::fopen(argv[1], "rb");
  if (!f || std::fread(img.data(), 1, img.size(), f) != img.size()) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
  std::fclose(f);
  cslam::ORBextractor ex(std::atoi(argv[4]), (float)std::atof(argv[5]), std::atoi(argv[6]), std::atoi(argv[7]), std::atoi(argv[8]));
  cv::Mat im(h, w, CV_8UC1, img.data(), (size_t)w);
  std::vector<cv::KeyPoint> keys;
  cv::Mat desc;
  ex(im, cv::Mat(), keys, desc);
  f = std::fopen(argv[9], "wb");
  const int n = (int)keys.size();
  std::fwrite(&n, 4, 1, f);
  for (int i = 0; i < n; i++) {
    const float v[5] = {keys[i].pt.x, keys[i].pt.y, keys[i].size, keys[i].angle, keys[i].response};
    std::fwrite(v, 4, 5, f);
    std::fwrite(&keys[i].octave, 4, 1, f);
  }
  for (int i = 0; i < n; i++) std::fwrite(desc.ptr(i), 1, 32, f);
  std::fclose(f);
  return 0;
}
