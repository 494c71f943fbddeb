// Host build of ccm_slam_amd/csrc/ba_math.h (tests/test_ba_math_host.py strips the HIP include and compiles this with g++ -ffp-contract=off):
// ba_oplus_fast against ba_oplus over 2 000 000 random updates, rotation angles 1e-5 .. 0.5 rad.
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include "ba_math_nohip.h"
int main() {
  srand(7);
  double worst_q = 0, worst_t = 0;
  for (int it = 0; it < 2000000; it++) {
    double u[6];
    const double mag = pow(10.0, -5.0 + 4.7 * (rand() / (double)RAND_MAX));   // theta scale 1e-5 .. 0.5
    for (int k = 0; k < 3; k++) u[k] = mag * (2.0 * rand() / RAND_MAX - 1.0) * 0.57;
    for (int k = 3; k < 6; k++) u[k] = 3.0 * (2.0 * rand() / RAND_MAX - 1.0);
    BaPose T = {0.1 * (rand() / (double)RAND_MAX), -0.3 * (rand() / (double)RAND_MAX), 0.2, 0.9, 5.0 * (rand() / (double)RAND_MAX - 0.5), 2.0, -7.0 * (rand() / (double)RAND_MAX)};
    ba_normalize_rotation(T);
    const BaPose a = ba_oplus(u, T), b = ba_oplus_fast(u, T);
    const double dq = fmax(fmax(fabs(a.qx - b.qx), fabs(a.qy - b.qy)), fmax(fabs(a.qz - b.qz), fabs(a.qw - b.qw)));
    const double dt = fmax(fmax(fabs(a.tx - b.tx), fabs(a.ty - b.ty)), fabs(a.tz - b.tz)) / (1.0 + fmax(fmax(fabs(a.tx), fabs(a.ty)), fabs(a.tz)));
    if (dq > worst_q) worst_q = dq;
    if (dt > worst_t) worst_t = dt;
  }
  printf("worst quaternion component difference %.3e, worst relative translation difference %.3e\n", worst_q, worst_t);
  return 0;
}
