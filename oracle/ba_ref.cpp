// ba_ref.cpp — CPU ORACLE (test infrastructure, NOT product code) for the Optimizer / g2o path.
//
// Restates on flat arrays what cslam::Optimizer builds and vendored g2o executes:
//   Optimizer::BundleAdjustmentClient / LocalBundleAdjustmentClient / MapFusionGBA
//       cslam/src/Optimizer.cpp:40-212, 349-644, 646-859
//   Optimizer::PoseOptimizationClient            cslam/src/Optimizer.cpp:215-347
//   g2o (paths under cslam/thirdparty/g2o/g2o/):
//     types/types_six_dof_expmap.{h,cpp}  EdgeSE3ProjectXYZ(+OnlyPose), VertexSE3Expmap
//     types/types_sba.h                    VertexSBAPointXYZ
//     types/se3quat.h, types/se3_ops.hpp   SE3Quat::exp, operator*, normalizeRotation
//     core/base_binary_edge.hpp:55-120, base_unary_edge.hpp:43-72   constructQuadraticForm
//     core/base_edge.h:58-61,96-102        chi2, robustInformation
//     core/robust_kernel_impl.cpp:78-90    RobustKernelHuber::robustify
//     core/block_solver.hpp:143-295,354-486,502-604   structure / Schur / setLambda
//     core/optimization_algorithm_levenberg.cpp:61-189  LM control
//     core/sparse_optimizer.cpp:61-114,199-267,354-435   active set, optimize loop, update
//     solvers/linear_solver_dense.h:65-113, linear_solver_eigen.h:106-136
// Eigen is not available here; its fixed-size primitives (Quaterniond(R), q*v, toRotationMatrix,
// 3x3 inverse, LDLT) are restated from their documented algorithms.
//
// PARITY PIN: the reference has no tests / golden vectors for this path and cannot be compiled in
// this container (Eigen absent) — "parity unpinned" against a running g2o.  The oracle is pinned
// instead by (i) an independent numpy/scipy derivation of one LM step (tests/test_oracle_ba.py),
// (ii) convergence to ground truth on noise-free scenes, (iii) dense-vs-sparse solver agreement.
//
// Stand-in: LinearSolverEigen = Eigen::SimplicialLDLT + AMD ordering.  Here: block (6x6) sparse
// Cholesky with a greedy minimum-degree ordering on the camera graph.  Same solution up to f64
// rounding; different elimination order, so last-bit differences vs Eigen are expected.
#include <cstdint>
#include <cmath>
#include <cstring>
#include <vector>
#include <map>
#include <set>
#include <algorithm>
#include <limits>
#include <chrono>
#include <cstdio>

namespace {

using std::vector;
typedef double M3[9];    // row-major 3x3
typedef double M6[36];   // row-major 6x6

struct Quat { double x, y, z, w; };
struct Pose { Quat q; double t[3]; };

inline double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Eigen::Quaternion::normalize + g2o SE3Quat::normalizeRotation (se3quat.h:280-285)
inline void normalize_rotation(Quat& q) {
  if (q.w < 0) { q.x = -q.x; q.y = -q.y; q.z = -q.z; q.w = -q.w; }
  const double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  q.x /= n; q.y /= n; q.z /= n; q.w /= n;
}
// Eigen quaternion product
inline Quat qmul(const Quat& a, const Quat& b) {
  Quat r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}
// Eigen QuaternionBase::_transformVector: v + w*uv + u x uv, uv = 2 (u x v)
inline void qrot(const Quat& q, const double v[3], double out[3]) {
  double uv[3] = {q.y * v[2] - q.z * v[1], q.z * v[0] - q.x * v[2], q.x * v[1] - q.y * v[0]};
  uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
  out[0] = v[0] + q.w * uv[0] + (q.y * uv[2] - q.z * uv[1]);
  out[1] = v[1] + q.w * uv[1] + (q.z * uv[0] - q.x * uv[2]);
  out[2] = v[2] + q.w * uv[2] + (q.x * uv[1] - q.y * uv[0]);
}
// Eigen QuaternionBase::toRotationMatrix
inline void qtoR(const Quat& q, M3 R) {
  const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
// Eigen Quaternion(Matrix3) (quaternionbase_assign_impl<Other,3,3>)
inline Quat RtoQ(const M3 m) {
  Quat q;
  double t = m[0] + m[4] + m[8];
  if (t > 0) {
    t = std::sqrt(t + 1.0);
    q.w = 0.5 * t;
    t = 0.5 / t;
    q.x = (m[7] - m[5]) * t; q.y = (m[2] - m[6]) * t; q.z = (m[3] - m[1]) * t;
  } else {
    int i = 0;
    if (m[4] > m[0]) i = 1;
    if (m[8] > m[i * 3 + i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0);
    double c[3];
    c[i] = 0.5 * t;
    t = 0.5 / t;
    q.w = (m[k * 3 + j] - m[j * 3 + k]) * t;
    c[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
    c[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
    q.x = c[0]; q.y = c[1]; q.z = c[2];
  }
  return q;
}
inline void pose_map(const Pose& T, const double X[3], double out[3]) {   // SE3Quat::map (se3quat.h:217-220)
  qrot(T.q, X, out);
  out[0] += T.t[0]; out[1] += T.t[1]; out[2] += T.t[2];
}
// SE3Quat::exp (se3quat.h:223-257); update = [omega(3), upsilon(3)]
inline Pose se3_exp(const double u[6]) {
  const double om[3] = {u[0], u[1], u[2]}, up[3] = {u[3], u[4], u[5]};
  const double theta = std::sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
  const M3 Om = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
  M3 Om2;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    double s = 0; for (int k = 0; k < 3; k++) s += Om[i * 3 + k] * Om[k * 3 + j];
    Om2[i * 3 + j] = s;
  }
  M3 R, V;
  if (theta < 0.00001) {
    for (int i = 0; i < 9; i++) { R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + Om[i] + Om2[i]; V[i] = R[i]; }
  } else {
    const double a = std::sin(theta) / theta, b = (1 - std::cos(theta)) / (theta * theta);
    const double c = (theta - std::sin(theta)) / std::pow(theta, 3);
    for (int i = 0; i < 9; i++) {
      const double I = (i % 4 == 0) ? 1.0 : 0.0;
      R[i] = I + a * Om[i] + b * Om2[i];
      V[i] = I + b * Om[i] + c * Om2[i];
    }
  }
  Pose P;
  P.q = RtoQ(R);
  for (int i = 0; i < 3; i++) P.t[i] = V[i * 3] * up[0] + V[i * 3 + 1] * up[1] + V[i * 3 + 2] * up[2];
  normalize_rotation(P.q);   // SE3Quat(const Quaterniond&, const Vector3d&) ctor (:61-63)
  return P;
}
// SE3Quat::operator* (se3quat.h:104-110)
inline Pose pose_mul(const Pose& a, const Pose& b) {
  Pose r = a;
  double rt[3];
  qrot(a.q, b.t, rt);
  r.t[0] += rt[0]; r.t[1] += rt[1]; r.t[2] += rt[2];
  r.q = qmul(a.q, b.q);
  normalize_rotation(r.q);
  return r;
}

// RobustKernelHuber::robustify (robust_kernel_impl.cpp:78-90)
// dsqr is a FLOAT member in the reference's g2o fork (robust_kernel_impl.h:84 `float dsqr;`, set by setDelta, .cpp:65-69): delta^2
// rounded to f32 is both the inlier threshold and the constant subtracted for outliers.  Found by pinning against oracle/_ref.
inline void huber(double e2, double delta, double rho[3]) {
  const double dsqr = (double)(float)(delta * delta);
  if (e2 <= dsqr) { rho[0] = e2; rho[1] = 1.; rho[2] = 0.; }
  else {
    const double sqrte = std::sqrt(e2);
    rho[0] = 2 * sqrte * delta - dsqr;
    rho[1] = delta / sqrte;
    rho[2] = -0.5 * rho[1] / e2;
  }
}

// 3x3 inverse by cofactors (Eigen fixed-size inverse, compute_inverse_size3_helper)
inline void inv3(const M3 m, M3 r) {
  const double c00 = m[4] * m[8] - m[5] * m[7], c10 = m[5] * m[6] - m[3] * m[8], c20 = m[3] * m[7] - m[4] * m[6];
  const double det = c00 * m[0] + c10 * m[1] + c20 * m[2];
  const double id = 1.0 / det;
  r[0] = c00 * id; r[3] = c10 * id; r[6] = c20 * id;
  r[1] = (m[2] * m[7] - m[1] * m[8]) * id; r[4] = (m[0] * m[8] - m[2] * m[6]) * id; r[7] = (m[1] * m[6] - m[0] * m[7]) * id;
  r[2] = (m[1] * m[5] - m[2] * m[4]) * id; r[5] = (m[2] * m[3] - m[0] * m[5]) * id; r[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

// ----- dense Cholesky (stand-in for Eigen::LDLT in LinearSolverDense) ---------------------------
bool dense_chol_solve(vector<double>& A, int n, const double* b, double* x) {
  // A row-major full symmetric; factor lower in place
  for (int j = 0; j < n; j++) {
    double d = A[(size_t)j * n + j];
    for (int k = 0; k < j; k++) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
    if (!(d > 0.0) || !std::isfinite(d)) return false;
    d = std::sqrt(d);
    A[(size_t)j * n + j] = d;
    for (int i = j + 1; i < n; i++) {
      double s = A[(size_t)i * n + j];
      const double* ri = &A[(size_t)i * n];
      const double* rj = &A[(size_t)j * n];
      for (int k = 0; k < j; k++) s -= ri[k] * rj[k];
      A[(size_t)i * n + j] = s / d;
    }
  }
  for (int i = 0; i < n; i++) {
    double s = b[i];
    for (int k = 0; k < i; k++) s -= A[(size_t)i * n + k] * x[k];
    x[i] = s / A[(size_t)i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double s = x[i];
    for (int k = i + 1; k < n; k++) s -= A[(size_t)k * n + i] * x[k];
    x[i] = s / A[(size_t)i * n + i];
  }
  return true;
}

// ----- block sparse Cholesky with greedy minimum-degree ordering -------------------------------
template <int BS>
struct BlockCholT {
  static constexpr int BB = BS * BS;
  int nb = 0;
  vector<int> perm, iperm;            // perm[k] = original block eliminated k-th
  vector<vector<int>> cstruct;        // per eliminated column k: sorted rows (positions > k)
  vector<vector<double>> Lcol;        // per column: (1 + |cstruct|) blocks of BB (diag first)
  bool analysed = false;

  void analyse(int n, const vector<std::pair<int, int>>& upper_blocks) {
    nb = n;
    vector<std::set<int>> adj(n);
    for (auto& p : upper_blocks) if (p.first != p.second) { adj[p.first].insert(p.second); adj[p.second].insert(p.first); }
    perm.assign(n, -1); iperm.assign(n, -1);
    vector<char> done(n, 0);
    std::set<std::pair<int, int>> heap;   // (degree, node)
    for (int i = 0; i < n; i++) heap.insert({(int)adj[i].size(), i});
    vector<vector<int>> nbrs(n);
    for (int k = 0; k < n; k++) {
      auto it = heap.begin();
      const int v = it->second;
      heap.erase(it);
      perm[k] = v; iperm[v] = k; done[v] = 1;
      vector<int> nb_v(adj[v].begin(), adj[v].end());
      nbrs[k] = nb_v;
      for (int a : nb_v) { heap.erase({(int)adj[a].size(), a}); adj[a].erase(v); }
      for (size_t i = 0; i < nb_v.size(); i++)
        for (size_t j = i + 1; j < nb_v.size(); j++) { adj[nb_v[i]].insert(nb_v[j]); adj[nb_v[j]].insert(nb_v[i]); }
      for (int a : nb_v) heap.insert({(int)adj[a].size(), a});
      adj[v].clear();
    }
    cstruct.assign(n, {});
    for (int k = 0; k < n; k++) {
      for (int a : nbrs[k]) cstruct[k].push_back(iperm[a]);
      std::sort(cstruct[k].begin(), cstruct[k].end());
    }
    Lcol.assign(n, {});
    for (int k = 0; k < n; k++) Lcol[k].assign((cstruct[k].size() + 1) * BB, 0.0);
    analysed = true;
  }

  static void cholB(double* A, bool& ok) {   // lower Cholesky in place (row-major), upper part ignored
    for (int j = 0; j < BS; j++) {
      double d = A[j * BS + j];
      for (int k = 0; k < j; k++) d -= A[j * BS + k] * A[j * BS + k];
      if (!(d > 0.0) || !std::isfinite(d)) { ok = false; return; }
      d = std::sqrt(d);
      A[j * BS + j] = d;
      for (int i = j + 1; i < BS; i++) {
        double s = A[i * BS + j];
        for (int k = 0; k < j; k++) s -= A[i * BS + k] * A[j * BS + k];
        A[i * BS + j] = s / d;
      }
    }
    for (int i = 0; i < BS; i++) for (int j = i + 1; j < BS; j++) A[i * BS + j] = 0.0;
  }

  // blocks: map (i<=j) -> 6x6 row-major (block (i,j) of the symmetric matrix, upper part)
  bool factor_solve(const vector<std::pair<int, int>>& keys, const vector<double>& vals, const double* b, double* x) {
    const int n = nb;
    for (int k = 0; k < n; k++) std::fill(Lcol[k].begin(), Lcol[k].end(), 0.0);
    // scatter A (lower form in permuted order): column = min(pos), row = max(pos)
    for (size_t kb = 0; kb < keys.size(); kb++) {
      const int i = keys[kb].first, j = keys[kb].second;
      const int pi = iperm[i], pj = iperm[j];
      const double* B = &vals[kb * BB];   // block (i,j)
      if (pi == pj) { std::memcpy(&Lcol[pi][0], B, BB * sizeof(double)); continue; }
      const int col = std::min(pi, pj), row = std::max(pi, pj);
      const auto& cs = cstruct[col];
      const int pos = (int)(std::lower_bound(cs.begin(), cs.end(), row) - cs.begin());
      double* dst = &Lcol[col][(size_t)(pos + 1) * BB];
      // need block (row_node, col_node) of A: if row corresponds to j (pj>pi) it is B^T, else B
      if (pj > pi) { for (int r = 0; r < BS; r++) for (int c = 0; c < BS; c++) dst[r * BS + c] = B[c * BS + r]; }
      else std::memcpy(dst, B, BB * sizeof(double));
    }
    bool ok = true;
    for (int k = 0; k < n && ok; k++) {
      double* D = &Lcol[k][0];
      cholB(D, ok);
      if (!ok) break;
      const auto& cs = cstruct[k];
      const int m = (int)cs.size();
      // L_rk = A_rk * L_kk^{-T}
      for (int r = 0; r < m; r++) {
        double* A = &Lcol[k][(size_t)(r + 1) * BB];
        for (int row = 0; row < BS; row++)
          for (int j = 0; j < BS; j++) {
            double s = A[row * BS + j];
            for (int p = 0; p < j; p++) s -= A[row * BS + p] * D[j * BS + p];
            A[row * BS + j] = s / D[j * BS + j];
          }
      }
      // trailing update
      for (int c = 0; c < m; c++) {
        const int colc = cs[c];
        const double* Lc = &Lcol[k][(size_t)(c + 1) * BB];
        const auto& cs2 = cstruct[colc];
        {  // diagonal block of column colc
          double* T = &Lcol[colc][0];
          for (int i = 0; i < BS; i++) for (int j = 0; j <= i; j++) {
            double s = 0; for (int p = 0; p < BS; p++) s += Lc[i * BS + p] * Lc[j * BS + p];
            T[i * BS + j] -= s;
          }
        }
        size_t pos = 0;
        for (int r = c + 1; r < m; r++) {
          const int rowr = cs[r];
          while (cs2[pos] < rowr) pos++;
          double* T = &Lcol[colc][(pos + 1) * BB];
          const double* Lr = &Lcol[k][(size_t)(r + 1) * BB];
          for (int i = 0; i < BS; i++) for (int j = 0; j < BS; j++) {
            double s = 0; for (int p = 0; p < BS; p++) s += Lr[i * BS + p] * Lc[j * BS + p];
            T[i * BS + j] -= s;
          }
        }
      }
    }
    if (!ok) return false;
    // solve L y = Pb ; L^T z = y
    vector<double> y((size_t)n * BS);
    for (int k = 0; k < n; k++) for (int i = 0; i < BS; i++) y[(size_t)k * BS + i] = b[(size_t)perm[k] * BS + i];
    for (int k = 0; k < n; k++) {
      const double* D = &Lcol[k][0];
      double* yk = &y[(size_t)k * BS];
      for (int i = 0; i < BS; i++) { double s = yk[i]; for (int p = 0; p < i; p++) s -= D[i * BS + p] * yk[p]; yk[i] = s / D[i * BS + i]; }
      const auto& cs = cstruct[k];
      for (size_t r = 0; r < cs.size(); r++) {
        const double* L = &Lcol[k][(r + 1) * BB];
        double* yr = &y[(size_t)cs[r] * BS];
        for (int i = 0; i < BS; i++) { double s = 0; for (int p = 0; p < BS; p++) s += L[i * BS + p] * yk[p]; yr[i] -= s; }
      }
    }
    for (int k = n - 1; k >= 0; k--) {
      const double* D = &Lcol[k][0];
      double* yk = &y[(size_t)k * BS];
      const auto& cs = cstruct[k];
      for (size_t r = 0; r < cs.size(); r++) {
        const double* L = &Lcol[k][(r + 1) * BB];
        const double* yr = &y[(size_t)cs[r] * BS];
        for (int p = 0; p < BS; p++) { double s = 0; for (int i = 0; i < BS; i++) s += L[i * BS + p] * yr[i]; yk[p] -= s; }
      }
      for (int i = (BS - 1); i >= 0; i--) { double s = yk[i]; for (int p = i + 1; p < BS; p++) s -= D[p * BS + i] * yk[p]; yk[i] = s / D[i * BS + i]; }
    }
    for (int k = 0; k < n; k++) for (int i = 0; i < BS; i++) x[(size_t)perm[k] * BS + i] = y[(size_t)k * BS + i];
    return true;
  }
};

typedef BlockCholT<6> BlockChol;

// ------------------------------------------------------------------------------------------------
struct Timers { double residuals = 0, quadratic = 0, schur = 0, linear = 0, update = 0, structure = 0; };

struct BA {
  int n_cam, n_pt, n_edge;
  vector<Pose> cam; vector<uint8_t> cam_fixed; const double* K;
  vector<double> pt;
  const int32_t *e_cam, *e_pt; const double *e_obs, *e_info; vector<uint8_t> e_level;
  double huber_delta;
  // active structure (g2o initializeOptimization + buildStructure)
  vector<int> act_edges;           // indices of active edges in insertion order
  vector<int> cam_slot;            // cam -> pose index (free cams with >=1 active edge), else -1
  vector<int> pt_slot;             // pt -> landmark index, else -1
  vector<int> slot_cam, slot_pt;
  int Cp = 0, Lp = 0;
  // system
  vector<double> Hpp;              // Cp*36 diagonal blocks (BA has no pose-pose edges)
  vector<double> Hll;              // Lp*9
  vector<double> Hpl;              // per active edge 6x3 (zero if cam fixed)
  vector<double> b;                // 6*Cp + 3*Lp
  vector<double> x;                // solution
  vector<vector<int>> pt_edges;    // per landmark slot: active-edge positions, ordered by pose slot (HplCCS column order)
  vector<double> err;              // per active edge 2
  // Schur pattern: upper-triangular 6x6 blocks (i<=j); blocks 0..Cp-1 are the diagonal ones
  vector<std::pair<int, int>> hs_keys; vector<double> hs_vals;
  vector<vector<int>> pt_pair_block;   // per landmark: block index for each (a, c2>=a) pair in elimination order
  BlockChol chol;
  int linear_solver = 0;           // 0 auto, 1 dense, 2 sparse
  Timers tm;
  int pt_begin = 0, pt_end = 0;    // landmark-slot shard [begin,end) used by the partial-system entry

  void project(const Pose& T, const double* Kc, const double X[3], double out[2], double* zc = nullptr) const {
    double Xc[3]; pose_map(T, X, Xc);
    out[0] = Xc[0] / Xc[2] * Kc[0] + Kc[2];   // cam_project (types_six_dof_expmap.cpp:141-147) + project2d
    out[1] = Xc[1] / Xc[2] * Kc[1] + Kc[3];
    if (zc) *zc = Xc[2];
  }

  void init_active(int level) {
    act_edges.clear();
    cam_slot.assign(n_cam, -1); pt_slot.assign(n_pt, -1);
    vector<char> cam_has(n_cam, 0), pt_has(n_pt, 0);
    for (int e = 0; e < n_edge; e++) {
      if (e_level[e] != level) continue;
      // allVerticesFixed never holds: points are never fixed in these graphs
      act_edges.push_back(e);
      cam_has[e_cam[e]] = 1; pt_has[e_pt[e]] = 1;
    }
    slot_cam.clear(); slot_pt.clear();
    for (int c = 0; c < n_cam; c++) if (cam_has[c] && !cam_fixed[c]) { cam_slot[c] = (int)slot_cam.size(); slot_cam.push_back(c); }
    for (int p = 0; p < n_pt; p++) if (pt_has[p]) { pt_slot[p] = (int)slot_pt.size(); slot_pt.push_back(p); }
    Cp = (int)slot_cam.size(); Lp = (int)slot_pt.size();
    pt_begin = 0; pt_end = Lp;
  }

  void build_structure() {
    const double t0 = now_ms();
    Hpp.assign((size_t)Cp * 36, 0.0); Hll.assign((size_t)Lp * 9, 0.0);
    Hpl.assign(act_edges.size() * 18, 0.0);
    b.assign((size_t)Cp * 6 + (size_t)Lp * 3, 0.0); x.assign(b.size(), 0.0);
    err.assign(act_edges.size() * 2, 0.0);
    pt_edges.assign(Lp, {});
    for (size_t k = 0; k < act_edges.size(); k++) {
      const int e = act_edges[k];
      if (cam_slot[e_cam[e]] >= 0) pt_edges[pt_slot[e_pt[e]]].push_back((int)k);
    }
    for (auto& v : pt_edges)
      std::stable_sort(v.begin(), v.end(), [&](int a, int c) { return cam_slot[e_cam[act_edges[a]]] < cam_slot[e_cam[act_edges[c]]]; });
    std::map<std::pair<int, int>, int> index;
    hs_keys.clear();
    for (int i = 0; i < Cp; i++) { index[{i, i}] = i; hs_keys.push_back({i, i}); }
    pt_pair_block.assign(Lp, {});
    for (int l = 0; l < Lp; l++) {
      const auto& v = pt_edges[l];
      for (size_t a = 0; a < v.size(); a++)
        for (size_t c = a; c < v.size(); c++) {
          const int i = cam_slot[e_cam[act_edges[v[a]]]], j = cam_slot[e_cam[act_edges[v[c]]]];
          auto key = std::make_pair(i, j);
          auto it = index.find(key);
          int id;
          if (it == index.end()) { id = (int)hs_keys.size(); index[key] = id; hs_keys.push_back(key); } else id = it->second;
          pt_pair_block[l].push_back(id);
        }
    }
    hs_vals.assign(hs_keys.size() * 36, 0.0);
    const bool sparse = (linear_solver == 2) || (linear_solver == 0 && Cp > 150);
    if (sparse) chol.analyse(Cp, hs_keys); else chol.analysed = false;
    tm.structure += now_ms() - t0;
  }

  // computeActiveErrors + activeRobustChi2 (sparse_optimizer.cpp:61-114)
  double compute_errors_chi2() {
    const double t0 = now_ms();
    double chi = 0;
    for (size_t k = 0; k < act_edges.size(); k++) {
      const int e = act_edges[k];
      double pr[2];
      project(cam[e_cam[e]], K + 4 * (size_t)e_cam[e], &pt[3 * (size_t)e_pt[e]], pr);
      const double e0 = e_obs[2 * (size_t)e] - pr[0], e1 = e_obs[2 * (size_t)e + 1] - pr[1];   // computeError (types_six_dof_expmap.h:90-95)
      err[2 * k] = e0; err[2 * k + 1] = e1;
      const double c2 = (e0 * e0 + e1 * e1) * e_info[e];   // chi2 = e^T Omega e, Omega = I*invSigma2
      if (huber_delta > 0) { double rho[3]; huber(c2, huber_delta, rho); chi += rho[0]; }
      else chi += c2;
    }
    tm.residuals += now_ms() - t0;
    return chi;
  }

  // BlockSolver::buildSystem (block_solver.hpp:502-560): linearizeOplus + constructQuadraticForm
  void build_system() {
    const double t0 = now_ms();
    std::fill(Hpp.begin(), Hpp.end(), 0.0); std::fill(Hll.begin(), Hll.end(), 0.0);
    std::fill(Hpl.begin(), Hpl.end(), 0.0); std::fill(b.begin(), b.end(), 0.0);
    for (size_t k = 0; k < act_edges.size(); k++) {
      const int e = act_edges[k];
      const int c = e_cam[e], p = e_pt[e];
      const double* Kc = K + 4 * (size_t)c;
      const double fx = Kc[0], fy = Kc[1];
      double Xc[3]; pose_map(cam[c], &pt[3 * (size_t)p], Xc);
      const double xx = Xc[0], y = Xc[1], z = Xc[2], z_2 = z * z;
      M3 R; qtoR(cam[c].q, R);
      // linearizeOplus (types_six_dof_expmap.cpp:103-139)
      const double tmp[6] = {fx, 0, -xx / z * fx, 0, fy, -y / z * fy};
      double Ji[6];   // 2x3  d e / d point
      for (int r = 0; r < 2; r++) for (int cc = 0; cc < 3; cc++) {
        double s = 0; for (int q = 0; q < 3; q++) s += tmp[r * 3 + q] * R[q * 3 + cc];
        Ji[r * 3 + cc] = -1. / z * s;
      }
      double Jj[12];  // 2x6  d e / d pose
      Jj[0] = xx * y / z_2 * fx; Jj[1] = -(1 + (xx * xx / z_2)) * fx; Jj[2] = y / z * fx; Jj[3] = -1. / z * fx; Jj[4] = 0; Jj[5] = xx / z_2 * fx;
      Jj[6] = (1 + y * y / z_2) * fy; Jj[7] = -xx * y / z_2 * fy; Jj[8] = -xx / z * fy; Jj[9] = 0; Jj[10] = -1. / z * fy; Jj[11] = y / z_2 * fy;
      // constructQuadraticForm (base_binary_edge.hpp:55-120)
      const double om = e_info[e];
      const double e0 = err[2 * k], e1 = err[2 * k + 1];
      double w = 1.0;
      if (huber_delta > 0) { double rho[3]; huber((e0 * e0 + e1 * e1) * om, huber_delta, rho); w = rho[1]; }
      const double omr0 = -om * e0 * w, omr1 = -om * e1 * w;   // omega_r = -Omega*e, scaled by rho[1]
      const double wom = w * om;                                 // robustInformation = rho[1]*Omega
      const int ls = pt_slot[p], cs = cam_slot[c];
      // from = vertex 0 = point (A = Ji), to = vertex 1 = pose (B = Jj)
      double* bl = &b[(size_t)Cp * 6 + (size_t)ls * 3];
      double* Hl = &Hll[(size_t)ls * 9];
      for (int i = 0; i < 3; i++) {
        bl[i] += Ji[i] * omr0 + Ji[3 + i] * omr1;
        for (int j = 0; j < 3; j++) Hl[i * 3 + j] += (Ji[i] * Ji[j] + Ji[3 + i] * Ji[3 + j]) * wom;
      }
      if (cs >= 0) {
        double* bp = &b[(size_t)cs * 6];
        double* Hp = &Hpp[(size_t)cs * 36];
        double* W = &Hpl[k * 18];
        for (int i = 0; i < 6; i++) {
          bp[i] += Jj[i] * omr0 + Jj[6 + i] * omr1;
          for (int j = 0; j < 6; j++) Hp[i * 6 + j] += (Jj[i] * Jj[j] + Jj[6 + i] * Jj[6 + j]) * wom;
          for (int j = 0; j < 3; j++) W[i * 3 + j] += (Jj[i] * Ji[j] + Jj[6 + i] * Ji[3 + j]) * wom;   // Hpl block (pose row, landmark col)
        }
      }
    }
    tm.quadratic += now_ms() - t0;
  }

  double max_diag() const {   // computeLambdaInit (optimization_algorithm_levenberg.cpp:166-180)
    double m = 0;
    for (int i = 0; i < Cp; i++) for (int j = 0; j < 6; j++) m = std::max(std::fabs(Hpp[(size_t)i * 36 + j * 7]), m);
    for (int i = 0; i < Lp; i++) for (int j = 0; j < 3; j++) m = std::max(std::fabs(Hll[(size_t)i * 9 + j * 4]), m);
    return m;
  }

  // Schur complement for landmark slots [lb,le) into Hschur / coeff; lambda added to both diagonals.
  // block_solver.hpp:367-439.  When add_hpp is false the Hpp (+lambda) term is left out (partial systems).
  void schur(double lambda, int lb, int le, bool add_hpp, bool skip_empty, vector<double>& coeff, vector<double>& Dinv_all, vector<double>& db_all) {
    std::fill(hs_vals.begin(), hs_vals.end(), 0.0);
    if (add_hpp)
      for (int i = 0; i < Cp; i++) {
        double* blk = &hs_vals[(size_t)i * 36];
        for (int q = 0; q < 36; q++) blk[q] = Hpp[(size_t)i * 36 + q];
        for (int q = 0; q < 6; q++) blk[q * 7] += lambda;
      }
    coeff.assign((size_t)Cp * 6, 0.0);
    Dinv_all.assign((size_t)Lp * 9, 0.0); db_all.assign((size_t)Lp * 3, 0.0);
    for (int l = lb; l < le; l++) {
      if (skip_empty && pt_edges[l].empty()) continue;
      size_t pair_pos = 0;
      M3 D, Dinv;
      for (int q = 0; q < 9; q++) D[q] = Hll[(size_t)l * 9 + q];
      D[0] += lambda; D[4] += lambda; D[8] += lambda;
      inv3(D, Dinv);
      std::memcpy(&Dinv_all[(size_t)l * 9], Dinv, sizeof(M3));
      const double* bl = &b[(size_t)Cp * 6 + (size_t)l * 3];
      double db[3];
      for (int i = 0; i < 3; i++) db[i] = Dinv[i * 3] * bl[0] + Dinv[i * 3 + 1] * bl[1] + Dinv[i * 3 + 2] * bl[2];
      std::memcpy(&db_all[(size_t)l * 3], db, sizeof(db));
      const auto& col = pt_edges[l];
      for (size_t a = 0; a < col.size(); a++) {
        const double* Bi = &Hpl[(size_t)col[a] * 18];
        const int i1 = cam_slot[e_cam[act_edges[col[a]]]];
        double BDinv[18];
        for (int r = 0; r < 6; r++) for (int c = 0; c < 3; c++)
          BDinv[r * 3 + c] = Bi[r * 3] * Dinv[c] + Bi[r * 3 + 1] * Dinv[3 + c] + Bi[r * 3 + 2] * Dinv[6 + c];
        for (int r = 0; r < 6; r++) coeff[(size_t)i1 * 6 + r] += Bi[r * 3] * db[0] + Bi[r * 3 + 1] * db[1] + Bi[r * 3 + 2] * db[2];
        for (size_t c2 = a; c2 < col.size(); c2++) {
          const double* Bj = &Hpl[(size_t)col[c2] * 18];
          const int i2 = cam_slot[e_cam[act_edges[col[c2]]]];
          (void)i2;
          double* H = &hs_vals[(size_t)pt_pair_block[l][pair_pos++] * 36];
          for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++)
            H[r * 6 + c] -= BDinv[r * 3] * Bj[c * 3] + BDinv[r * 3 + 1] * Bj[c * 3 + 1] + BDinv[r * 3 + 2] * Bj[c * 3 + 2];
        }
      }
    }
  }

  // BlockSolver::solve, Schur branch (block_solver.hpp:354-486)
  bool solve(double lambda) {
    double t0 = now_ms();
    vector<double> coeff, Dinv, db;
    schur(lambda, 0, Lp, true, false, coeff, Dinv, db);
    vector<double> bschur((size_t)Cp * 6);
    for (size_t i = 0; i < bschur.size(); i++) bschur[i] = b[i] - coeff[i];
    tm.schur += now_ms() - t0;
    t0 = now_ms();
    bool ok = true;
    if (Cp > 0) {
      if (chol.analysed) ok = chol.factor_solve(hs_keys, hs_vals, bschur.data(), x.data());
      else {
        const int n = Cp * 6;
        vector<double> A((size_t)n * n, 0.0);
        for (size_t kb = 0; kb < hs_keys.size(); kb++) {
          const int i = hs_keys[kb].first, j = hs_keys[kb].second;
          for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) {
            A[(size_t)(i * 6 + r) * n + j * 6 + c] = hs_vals[kb * 36 + r * 6 + c];
            A[(size_t)(j * 6 + c) * n + i * 6 + r] = hs_vals[kb * 36 + r * 6 + c];
          }
        }
        ok = dense_chol_solve(A, n, bschur.data(), x.data());
      }
    }
    tm.linear += now_ms() - t0;
    if (!ok) return false;
    // landmark back-substitution: xl = Dinv (bl - Hpl^T xp)   (:461-481)
    for (int l = 0; l < Lp; l++) {
      double cl[3] = {b[(size_t)Cp * 6 + (size_t)l * 3], b[(size_t)Cp * 6 + (size_t)l * 3 + 1], b[(size_t)Cp * 6 + (size_t)l * 3 + 2]};
      for (int k : pt_edges[l]) {
        const double* W = &Hpl[(size_t)k * 18];
        const double* xp = &x[(size_t)cam_slot[e_cam[act_edges[k]]] * 6];
        for (int c = 0; c < 3; c++) for (int r = 0; r < 6; r++) cl[c] -= W[r * 3 + c] * xp[r];
      }
      const double* Di = &Dinv[(size_t)l * 9];
      for (int i = 0; i < 3; i++) x[(size_t)Cp * 6 + (size_t)l * 3 + i] = Di[i * 3] * cl[0] + Di[i * 3 + 1] * cl[1] + Di[i * 3 + 2] * cl[2];
    }
    return true;
  }

  // SparseOptimizer::update (sparse_optimizer.cpp:422-435) + oplusImpl
  void apply_update() {
    const double t0 = now_ms();
    for (int i = 0; i < Cp; i++) {
      Pose& T = cam[slot_cam[i]];
      T = pose_mul(se3_exp(&x[(size_t)i * 6]), T);   // VertexSE3Expmap::oplusImpl (types_six_dof_expmap.h:73-76)
    }
    for (int l = 0; l < Lp; l++) {
      double* X = &pt[3 * (size_t)slot_pt[l]];
      for (int i = 0; i < 3; i++) X[i] += x[(size_t)Cp * 6 + (size_t)l * 3 + i];   // VertexSBAPointXYZ::oplusImpl (types_sba.h:52-56)
    }
    tm.update += now_ms() - t0;
  }
};

struct LMState { double lambda = 0, ni = 2; int nBad = 0; };

// test hook: called after every LM trial, right before the loop condition polls terminate() (levenberg.cpp:150) — lets a test
// raise the stop flag from another thread at a known trial on both the oracle and the product
typedef void (*ora_trial_cb)(void* user, int iteration, int trial_in_iteration, double chi2_trial, int accepted);
ora_trial_cb g_trial_cb = nullptr; void* g_trial_cb_user = nullptr;

// OptimizationAlgorithmLevenberg::solve (optimization_algorithm_levenberg.cpp:61-164); returns 0 OK, 1 Terminate
int lm_iteration(BA& ba, int iteration, LMState& st, const volatile unsigned char* stop, int* trials_out,
                 double* chi_out, double lambda_init_user, double* ini_chi_out = nullptr) {
  if (iteration == 0) ba.build_structure();
  double currentChi = ba.compute_errors_chi2();
  double tempChi = currentChi;
  const double iniChi = currentChi;
  if (ini_chi_out) *ini_chi_out = iniChi;
  ba.build_system();
  if (iteration == 0) {
    st.lambda = lambda_init_user > 0 ? lambda_init_user : 1e-5 * ba.max_diag();
    st.ni = 2; st.nBad = 0;
  }
  double rho = 0;
  int qmax = 0;
  do {
    const vector<Pose> cam_backup = ba.cam;       // push()
    const vector<double> pt_backup = ba.pt;
    const bool ok2 = ba.solve(st.lambda);
    ba.apply_update();
    tempChi = ba.compute_errors_chi2();
    if (!ok2) tempChi = std::numeric_limits<double>::max();
    rho = currentChi - tempChi;
    double scale = 0;                              // computeScale (:182-189)
    for (size_t j = 0; j < ba.x.size(); j++) scale += ba.x[j] * (st.lambda * ba.x[j] + ba.b[j]);
    scale += 1e-3;
    rho /= scale;
    if (rho > 0 && std::isfinite(tempChi)) {
      double alpha = 1. - std::pow((2 * rho - 1), 3);
      alpha = std::min(alpha, 2. / 3.);
      const double scaleFactor = std::max(1. / 3., alpha);
      st.lambda *= scaleFactor;
      st.ni = 2;
      currentChi = tempChi;
    } else {
      st.lambda *= st.ni;
      st.ni *= 2;
      ba.cam = cam_backup; ba.pt = pt_backup;      // pop()
    }
    qmax++;
    if (g_trial_cb) g_trial_cb(g_trial_cb_user, iteration, qmax, tempChi, (rho > 0 && std::isfinite(tempChi)) ? 1 : 0);
  } while (rho < 0 && qmax < 10 && !(stop && *stop));
  if (trials_out) *trials_out = qmax;
  if (chi_out) *chi_out = currentChi;
  if (qmax == 10 || rho == 0) return 1;
  if ((iniChi - currentChi) * 1e3 < iniChi) st.nBad++; else st.nBad = 0;
  if (st.nBad >= 3) return 1;
  return 0;
}

void load_problem(BA& ba, int n_cam, int n_pt, int n_edge, const double* cam_qt, const uint8_t* cam_fixed,
                  const double* cam_K, const double* pt_xyz, const int32_t* e_cam, const int32_t* e_pt,
                  const double* e_obs, const double* e_info, const uint8_t* e_level, double huber_delta) {
  ba.n_cam = n_cam; ba.n_pt = n_pt; ba.n_edge = n_edge;
  ba.cam.resize(n_cam);
  for (int c = 0; c < n_cam; c++) {
    const double* v = cam_qt + 7 * (size_t)c;
    ba.cam[c].q = {v[0], v[1], v[2], v[3]};
    ba.cam[c].t[0] = v[4]; ba.cam[c].t[1] = v[5]; ba.cam[c].t[2] = v[6];
    normalize_rotation(ba.cam[c].q);   // SE3Quat(q,t) ctor normalises (Converter::toSE3Quat, Converter.cc:40-50)
  }
  ba.cam_fixed.assign(cam_fixed, cam_fixed + n_cam);
  ba.K = cam_K;
  ba.pt.assign(pt_xyz, pt_xyz + 3 * (size_t)n_pt);
  ba.e_cam = e_cam; ba.e_pt = e_pt; ba.e_obs = e_obs; ba.e_info = e_info;
  ba.e_level.assign(n_edge, 0);
  if (e_level) ba.e_level.assign(e_level, e_level + n_edge);
  ba.huber_delta = huber_delta;
}

}  // namespace

extern "C" {

void ora_ba_set_trial_hook(ora_trial_cb cb, void* user) { g_trial_cb = cb; g_trial_cb_user = user; }

struct ora_ba_stats {
  int32_t iters_done, lm_trials, stop_reason;
  double chi2_initial, chi2_final, lambda_final;
  double ms_residuals, ms_quadratic, ms_schur, ms_linear, ms_update, ms_structure, ms_total;
  int32_t n_free_cams, n_active_pts, n_active_edges, n_schur_blocks;
  double chi2_hist[64]; double lambda_hist[64]; int32_t trials_hist[64];
};

// SparseOptimizer::optimize(n) (sparse_optimizer.cpp:354-419) on level-0 edges.
int ora_ba_optimize(int n_cam, int n_pt, int n_edge, double* cam_qt, const uint8_t* cam_fixed, const double* cam_K,
                    double* pt_xyz, const int32_t* e_cam, const int32_t* e_pt, const double* e_obs,
                    const double* e_info, const uint8_t* e_level, double huber_delta, int max_iters,
                    int linear_solver, double lambda_init, const volatile unsigned char* stop_flag,
                    double* chi2_per_edge, uint8_t* depth_pos, ora_ba_stats* stats) {
  const double t0 = now_ms();
  BA ba;
  load_problem(ba, n_cam, n_pt, n_edge, cam_qt, cam_fixed, cam_K, pt_xyz, e_cam, e_pt, e_obs, e_info, e_level, huber_delta);
  ba.linear_solver = linear_solver;
  ba.init_active(0);
  LMState st;
  int iters = 0, trials_total = 0, reason = 0;
  double chi_first = 0, chi_last = 0;
  if (stats) std::memset(stats, 0, sizeof(*stats));
  if (!ba.act_edges.empty()) {
    for (int i = 0; i < max_iters; i++) {
      if (stop_flag && *stop_flag) { reason = 1; break; }
      int trials = 0; double chi = 0, ini = 0;
      const int r = lm_iteration(ba, i, st, stop_flag, &trials, &chi, lambda_init, &ini);
      if (i == 0) chi_first = ini;
      trials_total += trials; iters++;
      chi_last = chi;
      if (stats && i < 64) { stats->chi2_hist[i] = chi; stats->lambda_hist[i] = st.lambda; stats->trials_hist[i] = trials; }
      if (r != 0) { reason = (st.nBad >= 3) ? 3 : 2; break; }
    }
  }
  // write back
  for (int c = 0; c < n_cam; c++) {
    double* v = cam_qt + 7 * (size_t)c;
    v[0] = ba.cam[c].q.x; v[1] = ba.cam[c].q.y; v[2] = ba.cam[c].q.z; v[3] = ba.cam[c].q.w;
    v[4] = ba.cam[c].t[0]; v[5] = ba.cam[c].t[1]; v[6] = ba.cam[c].t[2];
  }
  std::memcpy(pt_xyz, ba.pt.data(), sizeof(double) * 3 * (size_t)n_pt);
  // e->chi2() as the caller of optimize() sees it: _error holds the value of the LAST
  // computeActiveErrors (the last LM trial, even if that trial was rejected and popped); edges that
  // were not active keep whatever they had (entries left untouched).
  if (chi2_per_edge)
    for (size_t k = 0; k < ba.act_edges.size() && ba.err.size() == 2 * ba.act_edges.size(); k++) {
      const int e = ba.act_edges[k];
      chi2_per_edge[e] = (ba.err[2 * k] * ba.err[2 * k] + ba.err[2 * k + 1] * ba.err[2 * k + 1]) * e_info[e];
    }
  if (depth_pos)
    for (int e = 0; e < n_edge; e++) {
      double pr[2], zc;
      ba.project(ba.cam[e_cam[e]], cam_K + 4 * (size_t)e_cam[e], &ba.pt[3 * (size_t)e_pt[e]], pr, &zc);
      depth_pos[e] = zc > 0.0;   // isDepthPositive (types_six_dof_expmap.h:97-101), evaluated at the final estimate
    }
  if (stats) {
    stats->iters_done = iters; stats->lm_trials = trials_total; stats->stop_reason = reason;
    stats->chi2_initial = chi_first; stats->chi2_final = chi_last; stats->lambda_final = st.lambda;
    stats->ms_residuals = ba.tm.residuals; stats->ms_quadratic = ba.tm.quadratic; stats->ms_schur = ba.tm.schur;
    stats->ms_linear = ba.tm.linear; stats->ms_update = ba.tm.update; stats->ms_structure = ba.tm.structure;
    stats->ms_total = now_ms() - t0;
    stats->n_free_cams = ba.Cp; stats->n_active_pts = ba.Lp; stats->n_active_edges = (int)ba.act_edges.size();
    stats->n_schur_blocks = (int)ba.hs_keys.size();
  }
  return iters;
}

// robust chi2 of the active (level-0) edges at the given state
double ora_ba_chi2(int n_cam, int n_pt, int n_edge, const double* cam_qt, const double* cam_K, const double* pt_xyz,
                   const int32_t* e_cam, const int32_t* e_pt, const double* e_obs, const double* e_info,
                   const uint8_t* e_level, double huber_delta) {
  BA ba;
  vector<uint8_t> fixed(n_cam, 0);
  load_problem(ba, n_cam, n_pt, n_edge, cam_qt, fixed.data(), cam_K, pt_xyz, e_cam, e_pt, e_obs, e_info, e_level, huber_delta);
  ba.init_active(0);
  ba.err.assign(ba.act_edges.size() * 2, 0.0);
  return ba.compute_errors_chi2();
}

// Reduced camera system at the given state for the landmark-slot shard [shard, nshards):
// dense row-major Hschur (upper+lower filled) of size (6*Cp)^2, bschur, and the partial robust chi2.
// Hpp/b_p contributions come only from edges whose landmark is in the shard, so that the SUM over
// shards equals the full system (lambda is added on shard 0 only).  Used by the gloo tests that
// exercise the all-reduce algebra of the sharded global BA.
int ora_ba_partial_system(int n_cam, int n_pt, int n_edge, const double* cam_qt, const uint8_t* cam_fixed,
                          const double* cam_K, const double* pt_xyz, const int32_t* e_cam, const int32_t* e_pt,
                          const double* e_obs, const double* e_info, const uint8_t* e_level, double huber_delta,
                          double lambda, int pt_lo, int pt_hi /* original point index range */, int add_lambda,
                          double* Hs_dense, double* bs, double* chi2_partial, int* n_free_cams) {
  BA full;
  load_problem(full, n_cam, n_pt, n_edge, cam_qt, cam_fixed, cam_K, pt_xyz, e_cam, e_pt, e_obs, e_info, e_level, huber_delta);
  full.linear_solver = 1;
  full.init_active(0);
  const int Cp = full.Cp;
  if (n_free_cams) *n_free_cams = Cp;
  if (!Hs_dense) return 0;
  // restrict to the shard by deactivating edges whose point is outside [pt_lo, pt_hi) but keep slots of the full problem
  BA ba = full;
  vector<int> keep;
  for (int e : full.act_edges) if (e_pt[e] >= pt_lo && e_pt[e] < pt_hi) keep.push_back(e);
  ba.act_edges = keep;
  ba.build_structure();
  const double chi = ba.compute_errors_chi2();
  ba.build_system();
  vector<double> coeff_acc, Dinv, db;
  ba.schur(lambda, 0, ba.Lp, true, true, coeff_acc, Dinv, db);
  if (!add_lambda) for (int i = 0; i < Cp; i++) for (int q = 0; q < 6; q++) ba.hs_vals[(size_t)i * 36 + q * 7] -= lambda;
  const int n = Cp * 6;
  std::fill(Hs_dense, Hs_dense + (size_t)n * n, 0.0);
  for (size_t kb = 0; kb < ba.hs_keys.size(); kb++) {
    const int i = ba.hs_keys[kb].first, j = ba.hs_keys[kb].second;
    for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) {
      Hs_dense[(size_t)(i * 6 + r) * n + j * 6 + c] = ba.hs_vals[kb * 36 + r * 6 + c];
      Hs_dense[(size_t)(j * 6 + c) * n + i * 6 + r] = ba.hs_vals[kb * 36 + r * 6 + c];
    }
  }
  for (int i = 0; i < n; i++) bs[i] = ba.b[i] - coeff_acc[i];
  if (chi2_partial) *chi2_partial = chi;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// g2o::Sim3 (thirdparty/g2o/g2o/types/sim3.h).  The quaternion is NOT re-normalised by operator* or by the
// exponential-map constructor, exactly as upstream.
struct Sim3 { Quat r; double t[3]; double s; };

// Sim3(const Vector7d& update) (sim3.h:72-140); update = [omega(3), upsilon(3), sigma]
inline Sim3 sim3_exp(const double u[7]) {
  const double om[3] = {u[0], u[1], u[2]}, up[3] = {u[3], u[4], u[5]};
  const double sigma = u[6];
  const double theta = std::sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
  const M3 Om = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
  M3 Om2;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    double a = 0; for (int k = 0; k < 3; k++) a += Om[i * 3 + k] * Om[k * 3 + j];
    Om2[i * 3 + j] = a;
  }
  Sim3 S;
  S.s = std::exp(sigma);
  const double eps = 0.00001;
  double A, B, C;
  M3 R;
  auto small_R = [&]() { for (int i = 0; i < 9; i++) R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + Om[i] + Om2[i]; };
  auto full_R = [&]() {
    const double a = std::sin(theta) / theta, b = (1 - std::cos(theta)) / (theta * theta);
    for (int i = 0; i < 9; i++) R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + a * Om[i] + b * Om2[i];
  };
  if (std::fabs(sigma) < eps) {
    C = 1;
    if (theta < eps) { A = 1. / 2.; B = 1. / 6.; small_R(); }
    else {
      const double theta2 = theta * theta;
      A = (1 - std::cos(theta)) / theta2;
      B = (theta - std::sin(theta)) / (theta2 * theta);
      full_R();
    }
  } else {
    C = (S.s - 1) / sigma;
    if (theta < eps) {
      const double sigma2 = sigma * sigma;
      A = ((sigma - 1) * S.s + 1) / sigma2;
      B = ((0.5 * sigma2 - sigma + 1) * S.s) / (sigma2 * sigma);
      small_R();
    } else {
      full_R();
      const double a = S.s * std::sin(theta), b = S.s * std::cos(theta);
      const double theta2 = theta * theta, sigma2 = sigma * sigma;
      const double c = theta2 + sigma2;
      A = (a * sigma + (1 - b) * theta) / (theta * c);
      B = (C - ((b - 1) * sigma + a * theta) / (c)) * 1. / (theta2);
    }
  }
  S.r = RtoQ(R);
  for (int i = 0; i < 3; i++) {
    double acc = 0;
    for (int k = 0; k < 3; k++) {
      const double W = A * Om[i * 3 + k] + B * Om2[i * 3 + k] + C * ((i == k) ? 1.0 : 0.0);
      acc += W * up[k];
    }
    S.t[i] = acc;
  }
  return S;
}
inline Sim3 sim3_mul(const Sim3& a, const Sim3& b) {      // sim3.h:272-278
  Sim3 r;
  r.r = qmul(a.r, b.r);
  double rt[3]; qrot(a.r, b.t, rt);
  for (int i = 0; i < 3; i++) r.t[i] = a.s * rt[i] + a.t[i];
  r.s = a.s * b.s;
  return r;
}
inline Sim3 sim3_inv(const Sim3& a) {                     // sim3.h:240-243
  Sim3 r;
  r.r = {-a.r.x, -a.r.y, -a.r.z, a.r.w};
  const double k = -1. / a.s;
  const double v[3] = {k * a.t[0], k * a.t[1], k * a.t[2]};
  qrot(r.r, v, r.t);
  r.s = 1. / a.s;
  return r;
}
inline void sim3_map(const Sim3& S, const double X[3], double out[3]) {   // sim3.h:142-144
  double r[3]; qrot(S.r, X, r);
  for (int i = 0; i < 3; i++) out[i] = S.s * r[i] + S.t[i];
}

// Eigen::PartialPivLU<Matrix3d>(W).solve(t) (Sim3::log uses W.lu().solve(t), sim3.h:216): unblocked partial-pivot LU,
// row swap on the largest |entry| of the column, unit-lower forward and upper back substitution
inline void lu3_solve(const M3 Win, const double t[3], double x[3]) {
  double a[9]; for (int i = 0; i < 9; i++) a[i] = Win[i];
  int piv[3] = {0, 1, 2};
  for (int k = 0; k < 3; k++) {
    int best = k; double bv = std::fabs(a[k * 3 + k]);
    for (int r = k + 1; r < 3; r++) if (std::fabs(a[r * 3 + k]) > bv) { bv = std::fabs(a[r * 3 + k]); best = r; }
    if (best != k) { for (int c = 0; c < 3; c++) std::swap(a[k * 3 + c], a[best * 3 + c]); std::swap(piv[k], piv[best]); }
    for (int r = k + 1; r < 3; r++) a[r * 3 + k] /= a[k * 3 + k];
    for (int r = k + 1; r < 3; r++) for (int c = k + 1; c < 3; c++) a[r * 3 + c] -= a[r * 3 + k] * a[k * 3 + c];
  }
  double y[3] = {t[piv[0]], t[piv[1]], t[piv[2]]};
  y[1] -= a[3] * y[0];
  y[2] -= a[6] * y[0]; y[2] -= a[7] * y[1];
  x[2] = y[2] / a[8];
  x[1] = (y[1] - a[5] * x[2]) / a[4];
  x[0] = (y[0] - a[1] * x[1] - a[2] * x[2]) / a[0];
}

// Sim3::log (sim3.h:146-237) -> [omega(3), upsilon(3), sigma]
inline void sim3_log(const Sim3& S, double res[7]) {
  const double sigma = std::log(S.s);
  M3 R; qtoR(S.r, R);
  const double d = 0.5 * (R[0] + R[4] + R[8] - 1);
  const double dR[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};   // deltaR (se3_ops.hpp:40-47)
  double omega[3];
  const double eps = 0.00001;
  double A, B, C;
  auto small_angle = [&]() { for (int i = 0; i < 3; i++) omega[i] = 0.5 * dR[i]; };
  auto big_angle = [&](double theta) { const double k = theta / (2 * std::sqrt(1 - d * d)); for (int i = 0; i < 3; i++) omega[i] = k * dR[i]; };
  if (std::fabs(sigma) < eps) {
    C = 1;
    if (d > 1 - eps) { small_angle(); A = 1. / 2.; B = 1. / 6.; }
    else {
      const double theta = std::acos(d), theta2 = theta * theta;
      big_angle(theta);
      A = (1 - std::cos(theta)) / (theta2);
      B = (theta - std::sin(theta)) / (theta2 * theta);
    }
  } else {
    C = (S.s - 1) / sigma;
    if (d > 1 - eps) {
      const double sigma2 = sigma * sigma;
      small_angle();
      A = ((sigma - 1) * S.s + 1) / (sigma2);
      B = ((0.5 * sigma2 - sigma + 1) * S.s) / (sigma2 * sigma);
    } else {
      const double theta = std::acos(d);
      big_angle(theta);
      const double theta2 = theta * theta;
      const double a = S.s * std::sin(theta), b = S.s * std::cos(theta);
      const double c = theta2 + sigma * sigma;
      A = (a * sigma + (1 - b) * theta) / (theta * c);
      B = (C - ((b - 1) * sigma + a * theta) / (c)) * 1. / (theta2);
    }
  }
  const M3 Om = {0, -omega[2], omega[1], omega[2], 0, -omega[0], -omega[1], omega[0], 0};
  M3 Om2, W;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    double acc = 0; for (int k = 0; k < 3; k++) acc += Om[i * 3 + k] * Om[k * 3 + j];
    Om2[i * 3 + j] = acc;
  }
  for (int i = 0; i < 9; i++) W[i] = A * Om[i] + B * Om2[i] + C * ((i % 4 == 0) ? 1.0 : 0.0);
  double ups[3];
  lu3_solve(W, S.t, ups);
  for (int i = 0; i < 3; i++) { res[i] = omega[i]; res[i + 3] = ups[i]; }
  res[6] = sigma;
}

// ------------------------------------------------------------------------------------------------
// Optimizer::PoseOptimizationClient — cslam/src/Optimizer.cpp:215-347, flat restatement.
// One SE3 vertex, unary EdgeSE3ProjectXYZOnlyPose edges (types_six_dof_expmap.h:143-171,
// .cpp:266-288), Huber delta = (float)sqrt(5.991), 4 rounds x optimize(10) restarted from the input
// pose each round (:299), chi2 threshold 5.991f with chi2 cast to float (:318-320), kernel removed
// after the third round (:332-333), dense 6x6 solve (LinearSolverDense).  Returns
// nInitialCorrespondences - nBad (or 0 if fewer than 3 correspondences, :290-291).
int ora_pose_optimize(double* cam_qt, int n, const double* Xw, const double* obs, const double* info, const double* Kc,
                      uint8_t* outlier) {
  if (n < 3) return 0;
  Pose T0;
  T0.q = {cam_qt[0], cam_qt[1], cam_qt[2], cam_qt[3]};
  T0.t[0] = cam_qt[4]; T0.t[1] = cam_qt[5]; T0.t[2] = cam_qt[6];
  normalize_rotation(T0.q);
  Pose T = T0;
  const double delta = (double)(float)std::sqrt(5.991);   // const float deltaMono = sqrt(5.991) (:247)
  vector<uint8_t> level(n, 0), robust(n, 1);
  vector<double> err(2 * (size_t)n, 0.0);
  for (int i = 0; i < n; i++) outlier[i] = 0;
  const float chi2Mono[4] = {5.991f, 5.991f, 5.991f, 5.991f};
  int nBad = 0;
  auto compute_error = [&](int i) {
    double Xc[3]; pose_map(T, Xw + 3 * (size_t)i, Xc);
    err[2 * i] = obs[2 * i] - (Xc[0] / Xc[2] * Kc[0] + Kc[2]);
    err[2 * i + 1] = obs[2 * i + 1] - (Xc[1] / Xc[2] * Kc[1] + Kc[3]);
  };
  for (int it = 0; it < 4; it++) {
    T = T0;                                       // vSE3->setEstimate(Converter::toSE3Quat(Frame.mTcw)) (:299)
    vector<int> act;
    for (int i = 0; i < n; i++) if (level[i] == 0) act.push_back(i);
    if (!act.empty()) {                            // initializeOptimization(0) fails on an empty active set
      double lambda = 0, ni = 2; int nBadLM = 0;
      for (int iter = 0; iter < 10; iter++) {
        auto chi2_active = [&]() {
          double chi = 0;
          for (int i : act) {
            compute_error(i);
            const double c2 = (err[2 * i] * err[2 * i] + err[2 * i + 1] * err[2 * i + 1]) * info[i];
            if (robust[i]) { double rho[3]; huber(c2, delta, rho); chi += rho[0]; } else chi += c2;
          }
          return chi;
        };
        double currentChi = chi2_active();
        const double iniChi = currentChi;
        double H[36] = {0}, b[6] = {0};
        for (int i : act) {
          double Xc[3]; pose_map(T, Xw + 3 * (size_t)i, Xc);
          const double x = Xc[0], y = Xc[1], invz = 1.0 / Xc[2], invz_2 = invz * invz;
          const double fx = Kc[0], fy = Kc[1];
          double J[12];   // EdgeSE3ProjectXYZOnlyPose::linearizeOplus (types_six_dof_expmap.cpp:266-288)
          J[0] = x * y * invz_2 * fx; J[1] = -(1 + (x * x * invz_2)) * fx; J[2] = y * invz * fx; J[3] = -invz * fx; J[4] = 0; J[5] = x * invz_2 * fx;
          J[6] = (1 + y * y * invz_2) * fy; J[7] = -x * y * invz_2 * fy; J[8] = -x * invz * fy; J[9] = 0; J[10] = -invz * fy; J[11] = y * invz_2 * fy;
          const double om = info[i];
          double w = 1.0;
          if (robust[i]) { double rho[3]; huber((err[2 * i] * err[2 * i] + err[2 * i + 1] * err[2 * i + 1]) * om, delta, rho); w = rho[1]; }
          const double o0 = -om * err[2 * i] * w, o1 = -om * err[2 * i + 1] * w, wom = w * om;   // base_unary_edge.hpp:43-72
          for (int r = 0; r < 6; r++) {
            b[r] += J[r] * o0 + J[6 + r] * o1;
            for (int c = 0; c < 6; c++) H[r * 6 + c] += (J[r] * J[c] + J[6 + r] * J[6 + c]) * wom;
          }
        }
        if (iter == 0) { double m = 0; for (int j = 0; j < 6; j++) m = std::max(std::fabs(H[j * 7]), m); lambda = 1e-5 * m; ni = 2; nBadLM = 0; }
        double rho = 0, tempChi; int qmax = 0; double xs[6];
        do {
          const Pose backup = T;
          vector<double> A(36);
          for (int q = 0; q < 36; q++) A[q] = H[q];
          for (int q = 0; q < 6; q++) A[q * 7] += lambda;
          const bool ok2 = dense_chol_solve(A, 6, b, xs);
          T = pose_mul(se3_exp(xs), T);
          tempChi = chi2_active();
          if (!ok2) tempChi = std::numeric_limits<double>::max();
          rho = currentChi - tempChi;
          double scale = 0;
          for (int j = 0; j < 6; j++) scale += xs[j] * (lambda * xs[j] + b[j]);
          scale += 1e-3;
          rho /= scale;
          if (rho > 0 && std::isfinite(tempChi)) {
            double alpha = 1. - std::pow((2 * rho - 1), 3);
            alpha = std::min(alpha, 2. / 3.);
            lambda *= std::max(1. / 3., alpha);
            ni = 2; currentChi = tempChi;
          } else { lambda *= ni; ni *= 2; T = backup; }
          qmax++;
        } while (rho < 0 && qmax < 10);
        if (qmax == 10 || rho == 0) break;
        if ((iniChi - currentChi) * 1e3 < iniChi) nBadLM++; else nBadLM = 0;
        if (nBadLM >= 3) break;
      }
    }
    nBad = 0;
    for (int i = 0; i < n; i++) {
      if (outlier[i]) compute_error(i);             // (:313-316); active edges keep the _error of the LAST evaluated
                                                    // trial (g2o does not recompute after a rejected step)
      const float chi2 = (float)((err[2 * i] * err[2 * i] + err[2 * i + 1] * err[2 * i + 1]) * info[i]);
      if (chi2 > chi2Mono[it]) { outlier[i] = 1; level[i] = 1; nBad++; }
      else { outlier[i] = 0; level[i] = 0; }
      if (it == 2) robust[i] = 0;
    }
    if (n < 10) break;                              // optimizer.edges().size() < 10 (:336-337)
  }
  cam_qt[0] = T.q.x; cam_qt[1] = T.q.y; cam_qt[2] = T.q.z; cam_qt[3] = T.q.w;
  cam_qt[4] = T.t[0]; cam_qt[5] = T.t[1]; cam_qt[6] = T.t[2];
  return n - nBad;
}


// ------------------------------------------------------------------------------------------------
// Optimizer::OptimizeSim3 — cslam/src/Optimizer.cpp:861-1056, flat restatement.
// One VertexSim3Expmap (oplus: Sim3(update) * estimate, update[6] zeroed when _fix_scale,
// types_seven_dof_expmap.h:58-67); per correspondence i two edges with FIXED point vertices:
//   e12_i (EdgeSim3ProjectXYZ, :133-151):        obs1 - cam_map1(project(S12.map(P2c)))
//   e21_i (EdgeInverseSim3ProjectXYZ, :154-172): obs2 - cam_map2(project(S12.inverse().map(P1c)))
// info = invSigma2 * I, Huber delta = (float)sqrt(th2).  Neither edge overrides linearizeOplus, so g2o
// differentiates NUMERICALLY (base_binary_edge.hpp:129-196: central differences, delta 1e-9, through the
// vertex oplus) — restated literally because it defines the reference's Jacobian to ~1e-7.
// BlockSolverX + LinearSolverDense (7x7), Levenberg.  optimize(5); drop pairs with chi2 > th2 on either edge;
// optimize(10 if any dropped else 5) on the rest; count inliers.  Returns 0 and leaves sim3 untouched when
// fewer than 10 pairs survive the first pass (:1015-1016).
// sim3 = [qx qy qz qw tx ty tz s];  inlier[i] = 1 while vpMatches1[idx] stays non-null.
int ora_sim3_optimize(double* sim3, int n, const double* P1c, const double* P2c, const double* obs1, const double* obs2,
                      const double* info1, const double* info2, const double* K1, const double* K2, double th2, int fix_scale,
                      uint8_t* inlier) {
  Sim3 S;
  S.r = {sim3[0], sim3[1], sim3[2], sim3[3]};
  S.t[0] = sim3[4]; S.t[1] = sim3[5]; S.t[2] = sim3[6]; S.s = sim3[7];
  const double delta = (double)(float)std::sqrt((float)th2);   // const float deltaHuber = sqrt(th2), th2 is float (:908)
  vector<uint8_t> alive(n, 1);
  vector<double> err(4 * (size_t)n, 0.0);   // [e12 (2) | e21 (2)] per pair
  for (int i = 0; i < n; i++) inlier[i] = 1;
  auto err12 = [&](const Sim3& X, int i, double e[2]) {
    double p[3]; sim3_map(X, P2c + 3 * (size_t)i, p);
    e[0] = obs1[2 * i] - ((p[0] / p[2]) * K1[0] + K1[2]);
    e[1] = obs1[2 * i + 1] - ((p[1] / p[2]) * K1[1] + K1[3]);
  };
  auto err21 = [&](const Sim3& Xinv, int i, double e[2]) {
    double p[3]; sim3_map(Xinv, P1c + 3 * (size_t)i, p);
    e[0] = obs2[2 * i] - ((p[0] / p[2]) * K2[0] + K2[2]);
    e[1] = obs2[2 * i + 1] - ((p[1] / p[2]) * K2[1] + K2[3]);
  };
  auto oplus = [&](const Sim3& X, const double* upd) {
    double u[7]; for (int k = 0; k < 7; k++) u[k] = upd[k];
    if (fix_scale) u[6] = 0;
    return sim3_mul(sim3_exp(u), X);
  };
  auto chi2_of = [&](const double* e, double om) { return (e[0] * e[0] + e[1] * e[1]) * om; };
  auto optimize = [&](int iters) {
    vector<int> act;
    for (int i = 0; i < n; i++) if (alive[i]) act.push_back(i);
    if (act.empty()) return;
    double lambda = 0, ni = 2; int nBadLM = 0;
    auto chi2_active = [&]() {
      const Sim3 Sinv = sim3_inv(S);
      double chi = 0;
      for (int i : act) {
        err12(S, i, &err[4 * (size_t)i]);
        err21(Sinv, i, &err[4 * (size_t)i + 2]);
        double rho[3];
        huber(chi2_of(&err[4 * (size_t)i], info1[i]), delta, rho); chi += rho[0];
        huber(chi2_of(&err[4 * (size_t)i + 2], info2[i]), delta, rho); chi += rho[0];
      }
      return chi;
    };
    for (int iter = 0; iter < iters; iter++) {
      double currentChi = chi2_active();
      const double iniChi = currentChi;
      // numeric Jacobians: the 14 perturbed estimates are the same for every edge
      Sim3 Sp[7], Sm[7], Spi[7], Smi[7];
      const double dlt = 1e-9, scalar = 1.0 / (2 * dlt);
      for (int d = 0; d < 7; d++) {
        double add[7] = {0, 0, 0, 0, 0, 0, 0};
        add[d] = dlt;  Sp[d] = oplus(S, add); Spi[d] = sim3_inv(Sp[d]);
        add[d] = -dlt; Sm[d] = oplus(S, add); Smi[d] = sim3_inv(Sm[d]);
      }
      double H[49] = {0}, b[7] = {0};
      auto add_edge = [&](const double J[14], const double* e, double om) {
        double rho[3]; huber(chi2_of(e, om), delta, rho);
        const double w = rho[1];
        const double o0 = -om * e[0] * w, o1 = -om * e[1] * w, wom = w * om;
        for (int r = 0; r < 7; r++) {
          b[r] += J[r] * o0 + J[7 + r] * o1;
          for (int c = 0; c < 7; c++) H[r * 7 + c] += (J[r] * wom) * J[c] + (J[7 + r] * wom) * J[7 + c];
        }
      };
      for (int i : act) {
        double J[14];
        for (int d = 0; d < 7; d++) {
          double ep[2], em[2];
          err12(Sp[d], i, ep); err12(Sm[d], i, em);
          J[d] = scalar * (ep[0] - em[0]); J[7 + d] = scalar * (ep[1] - em[1]);
        }
        add_edge(J, &err[4 * (size_t)i], info1[i]);
        for (int d = 0; d < 7; d++) {
          double ep[2], em[2];
          err21(Spi[d], i, ep); err21(Smi[d], i, em);
          J[d] = scalar * (ep[0] - em[0]); J[7 + d] = scalar * (ep[1] - em[1]);
        }
        add_edge(J, &err[4 * (size_t)i + 2], info2[i]);
      }
      if (iter == 0) { double m = 0; for (int j = 0; j < 7; j++) m = std::max(std::fabs(H[j * 8]), m); lambda = 1e-5 * m; ni = 2; nBadLM = 0; }
      double rho = 0, tempChi; int qmax = 0; double xs[7];
      do {
        const Sim3 backup = S;
        vector<double> A(49);
        for (int q = 0; q < 49; q++) A[q] = H[q];
        for (int q = 0; q < 7; q++) A[q * 8] += lambda;
        const bool ok2 = dense_chol_solve(A, 7, b, xs);
        if (!ok2) for (int q = 0; q < 7; q++) xs[q] = 0;
        S = oplus(S, xs);
        tempChi = chi2_active();
        if (!ok2) tempChi = std::numeric_limits<double>::max();
        rho = currentChi - tempChi;
        double scale = 0;
        for (int j = 0; j < 7; j++) scale += xs[j] * (lambda * xs[j] + b[j]);
        scale += 1e-3;
        rho /= scale;
        if (rho > 0 && std::isfinite(tempChi)) {
          double alpha = 1. - std::pow((2 * rho - 1), 3);
          alpha = std::min(alpha, 2. / 3.);
          lambda *= std::max(1. / 3., alpha);
          ni = 2; currentChi = tempChi;
        } else { lambda *= ni; ni *= 2; S = backup; }
        qmax++;
      } while (rho < 0 && qmax < 10);
      if (qmax == 10 || rho == 0) break;
      if ((iniChi - currentChi) * 1e3 < iniChi) nBadLM++; else nBadLM = 0;
      if (nBadLM >= 3) break;
    }
  };
  optimize(5);
  int nBad = 0;
  for (int i = 0; i < n; i++)   // e->chi2() reads the _error of the last evaluated trial (no recompute after a rejected step)
    if (chi2_of(&err[4 * (size_t)i], info1[i]) > th2 || chi2_of(&err[4 * (size_t)i + 2], info2[i]) > th2) { alive[i] = 0; inlier[i] = 0; nBad++; }
  const int more = nBad > 0 ? 10 : 5;
  if (n - nBad < 10) return 0;
  optimize(more);
  int nIn = 0;
  for (int i = 0; i < n; i++) {
    if (!alive[i]) continue;
    if (chi2_of(&err[4 * (size_t)i], info1[i]) > th2 || chi2_of(&err[4 * (size_t)i + 2], info2[i]) > th2) inlier[i] = 0;
    else nIn++;
  }
  sim3[0] = S.r.x; sim3[1] = S.r.y; sim3[2] = S.r.z; sim3[3] = S.r.w;
  sim3[4] = S.t[0]; sim3[5] = S.t[1]; sim3[6] = S.t[2]; sim3[7] = S.s;
  return nIn;
}


// ------------------------------------------------------------------------------------------------
// Optimizer::OptimizeEssentialGraphLoopClosure / MapFusion numerics — cslam/src/Optimizer.cpp:1058-1331, 1333-1566.
// Flat restatement of what both functions hand to g2o: VertexSim3Expmap per keyframe (estimate Siw, one vertex fixed,
// _fix_scale), EdgeSim3 per spanning-tree / loop / covisibility link with vertex(0) = i, vertex(1) = j, measurement
// Sji and information I7 (types_seven_dof_expmap.h:98-121: error = log(Sji * Siw * Sjw^-1)); BlockSolver_7_3 with
// LinearSolverEigen, Levenberg with setUserLambdaInit(1e-16), optimize(20).  EdgeSim3 does not override
// linearizeOplus, so both 7x7 Jacobians come from g2o's numeric differentiation (base_binary_edge.hpp:129-196).
// The graph walking that chooses the edges and the SE3 / map-point write-back (:1268-1330) stay with the caller.
struct ora_pg_stats { int32_t iters_done, lm_trials; double chi2_initial, chi2_final, lambda_final; };

int ora_pose_graph_optimize(int n_vert, double* sim3, const uint8_t* fixed, int fix_scale, int n_edge, const int32_t* e_i, const int32_t* e_j,
                            const double* meas, int max_iters, double lambda_init, ora_pg_stats* stats) {
  vector<Sim3> S(n_vert), C(n_edge);
  auto load = [](const double* p) { Sim3 x; x.r = {p[0], p[1], p[2], p[3]}; x.t[0] = p[4]; x.t[1] = p[5]; x.t[2] = p[6]; x.s = p[7]; return x; };
  for (int v = 0; v < n_vert; v++) S[v] = load(sim3 + 8 * (size_t)v);
  for (int e = 0; e < n_edge; e++) C[e] = load(meas + 8 * (size_t)e);
  // index mapping: non-fixed vertices in id order (sparse_optimizer.cpp:166-190)
  vector<int> slot(n_vert, -1); int nfree = 0;
  for (int v = 0; v < n_vert; v++) if (!fixed[v]) slot[v] = nfree++;
  // active edges: not both vertices fixed
  vector<int> act;
  for (int e = 0; e < n_edge; e++) if (!(fixed[e_i[e]] && fixed[e_j[e]])) act.push_back(e);
  ora_pg_stats st{}; st.iters_done = 0; st.lm_trials = 0;
  if (nfree == 0 || act.empty()) { if (stats) *stats = st; return 0; }
  auto oplus = [&](const Sim3& X, const double* upd) {
    double u[7]; for (int k = 0; k < 7; k++) u[k] = upd[k];
    if (fix_scale) u[6] = 0;
    return sim3_mul(sim3_exp(u), X);
  };
  auto edge_error = [&](int e, const Sim3& Si, const Sim3& Sj, double err[7]) {
    const Sim3 E = sim3_mul(sim3_mul(C[e], Si), sim3_inv(Sj));
    sim3_log(E, err);
  };
  vector<double> err(7 * (size_t)n_edge, 0.0);
  auto chi2_active = [&]() {
    double chi = 0;
    for (int e : act) {
      double* er = &err[7 * (size_t)e];
      edge_error(e, S[e_i[e]], S[e_j[e]], er);
      double c2 = 0; for (int k = 0; k < 7; k++) c2 += er[k] * er[k];
      chi += c2;
    }
    return chi;
  };
  // block structure: diagonal + one upper block per connected pair of free vertices
  std::map<std::pair<int, int>, int> blk;
  vector<std::pair<int, int>> keys;
  for (int v = 0; v < nfree; v++) { blk[{v, v}] = (int)keys.size(); keys.push_back({v, v}); }
  for (int e : act) {
    const int a = slot[e_i[e]], b = slot[e_j[e]];
    if (a < 0 || b < 0 || a == b) continue;
    const std::pair<int, int> k{std::min(a, b), std::max(a, b)};
    if (!blk.count(k)) { blk[k] = (int)keys.size(); keys.push_back(k); }
  }
  BlockCholT<7> chol;
  chol.analyse(nfree, keys);
  vector<double> H(49 * keys.size()), bvec(7 * (size_t)nfree), x(7 * (size_t)nfree);
  double lambda = lambda_init, ni = 2; int nBad = 0;
  double currentChi = 0;
  for (int iter = 0; iter < max_iters; iter++) {
    currentChi = chi2_active();
    if (iter == 0) st.chi2_initial = currentChi;
    const double iniChi = currentChi;
    std::fill(H.begin(), H.end(), 0.0); std::fill(bvec.begin(), bvec.end(), 0.0);
    const double dlt = 1e-9, scalar = 1.0 / (2 * dlt);
    for (int e : act) {
      const int vi = e_i[e], vj = e_j[e];
      const int a = slot[vi], b = slot[vj];
      double Ji[49], Jj[49];   // [row k of the error][column d]
      for (int side = 0; side < 2; side++) {
        if ((side == 0 ? a : b) < 0) continue;
        double* J = side == 0 ? Ji : Jj;
        for (int d = 0; d < 7; d++) {
          double add[7] = {0, 0, 0, 0, 0, 0, 0}, ep[7], em[7];
          add[d] = dlt;
          if (side == 0) edge_error(e, oplus(S[vi], add), S[vj], ep); else edge_error(e, S[vi], oplus(S[vj], add), ep);
          add[d] = -dlt;
          if (side == 0) edge_error(e, oplus(S[vi], add), S[vj], em); else edge_error(e, S[vi], oplus(S[vj], add), em);
          for (int k = 0; k < 7; k++) J[k * 7 + d] = scalar * (ep[k] - em[k]);
        }
      }
      const double* er = &err[7 * (size_t)e];
      auto addJtJ = [&](int r, int c, const double* A, const double* B, bool transpose_store) {
        double* Hb = &H[49 * (size_t)blk[{std::min(r, c), std::max(r, c)}]];
        for (int p = 0; p < 7; p++) for (int q = 0; q < 7; q++) {
          double acc = 0; for (int k = 0; k < 7; k++) acc += A[k * 7 + p] * B[k * 7 + q];
          if (transpose_store) Hb[q * 7 + p] += acc; else Hb[p * 7 + q] += acc;
        }
      };
      if (a >= 0) {
        for (int p = 0; p < 7; p++) { double acc = 0; for (int k = 0; k < 7; k++) acc += Ji[k * 7 + p] * (-er[k]); bvec[7 * (size_t)a + p] += acc; }
        addJtJ(a, a, Ji, Ji, false);
        if (b >= 0 && a != b) addJtJ(a, b, Ji, Jj, a > b);   // block (a,b) = Ji^T Jj; stored in the upper block (min,max)
      }
      if (b >= 0) {
        for (int p = 0; p < 7; p++) { double acc = 0; for (int k = 0; k < 7; k++) acc += Jj[k * 7 + p] * (-er[k]); bvec[7 * (size_t)b + p] += acc; }
        addJtJ(b, b, Jj, Jj, false);
      }
    }
    if (iter == 0 && !(lambda_init > 0)) {   // computeLambdaInit (levenberg.cpp:167-181) when no user value is set
      double m = 0;
      for (int v = 0; v < nfree; v++) for (int k = 0; k < 7; k++) m = std::max(m, std::fabs(H[49 * (size_t)v + k * 8]));
      lambda = 1e-5 * m;
    }
    double rho = 0, tempChi; int qmax = 0;
    do {
      vector<Sim3> backup = S;
      vector<double> Hd = H;
      for (int v = 0; v < nfree; v++) for (int k = 0; k < 7; k++) Hd[49 * (size_t)v + k * 8] += lambda;
      const bool ok2 = chol.factor_solve(keys, Hd, bvec.data(), x.data());
      if (!ok2) std::fill(x.begin(), x.end(), 0.0);
      for (int v = 0; v < n_vert; v++) if (slot[v] >= 0) S[v] = oplus(S[v], &x[7 * (size_t)slot[v]]);
      tempChi = chi2_active();
      st.lm_trials++;
      if (!ok2) tempChi = std::numeric_limits<double>::max();
      rho = currentChi - tempChi;
      double scale = 0;
      for (size_t j = 0; j < x.size(); j++) scale += x[j] * (lambda * x[j] + bvec[j]);
      scale += 1e-3;
      rho /= scale;
      if (rho > 0 && std::isfinite(tempChi)) {
        double alpha = 1. - std::pow((2 * rho - 1), 3);
        alpha = std::min(alpha, 2. / 3.);
        lambda *= std::max(1. / 3., alpha);
        ni = 2; currentChi = tempChi;
      } else { lambda *= ni; ni *= 2; S = backup; }
      qmax++;
    } while (rho < 0 && qmax < 10);
    st.iters_done++;
    if (qmax == 10 || rho == 0) break;
    if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
    if (nBad >= 3) break;
  }
  st.chi2_final = currentChi; st.lambda_final = lambda;
  for (int v = 0; v < n_vert; v++) {
    double* p = sim3 + 8 * (size_t)v;
    p[0] = S[v].r.x; p[1] = S[v].r.y; p[2] = S[v].r.z; p[3] = S[v].r.w; p[4] = S[v].t[0]; p[5] = S[v].t[1]; p[6] = S[v].t[2]; p[7] = S[v].s;
  }
  if (stats) *stats = st;
  return st.iters_done;
}


// Debug export for preconditioner studies: dense H (7F x 7F, row-major, no lambda), b, and the numeric Jacobians of every
// edge ([E][2][49], zero for fixed sides) at the given state.  Free slots in vertex order.
int ora_pose_graph_system(int n_vert, const double* sim3, const uint8_t* fixed, int fix_scale, int n_edge, const int32_t* e_i,
                          const int32_t* e_j, const double* meas, double* Hdense, double* bout, double* Jout) {
  vector<Sim3> S(n_vert), C(n_edge);
  auto load = [](const double* p) { Sim3 x; x.r = {p[0], p[1], p[2], p[3]}; x.t[0] = p[4]; x.t[1] = p[5]; x.t[2] = p[6]; x.s = p[7]; return x; };
  for (int v = 0; v < n_vert; v++) S[v] = load(sim3 + 8 * (size_t)v);
  for (int e = 0; e < n_edge; e++) C[e] = load(meas + 8 * (size_t)e);
  vector<int> slot(n_vert, -1); int F = 0;
  for (int v = 0; v < n_vert; v++) if (!fixed[v]) slot[v] = F++;
  const size_t n = 7 * (size_t)F;
  std::fill(Hdense, Hdense + n * n, 0.0); std::fill(bout, bout + n, 0.0); std::fill(Jout, Jout + 98 * (size_t)n_edge, 0.0);
  auto oplus = [&](const Sim3& X, const double* upd) { double u[7]; for (int k = 0; k < 7; k++) u[k] = upd[k]; if (fix_scale) u[6] = 0; return sim3_mul(sim3_exp(u), X); };
  auto edge_error = [&](int e, const Sim3& Si, const Sim3& Sj, double err[7]) { sim3_log(sim3_mul(sim3_mul(C[e], Si), sim3_inv(Sj)), err); };
  const double dlt = 1e-9, scalar = 1.0 / (2 * dlt);
  for (int e = 0; e < n_edge; e++) {
    const int vi = e_i[e], vj = e_j[e], a = slot[vi], b = slot[vj];
    if (a < 0 && b < 0) continue;
    double er[7]; edge_error(e, S[vi], S[vj], er);
    double* Ji = Jout + 98 * (size_t)e; double* Jj = Ji + 49;
    for (int side = 0; side < 2; side++) {
      if ((side == 0 ? a : b) < 0) continue;
      double* J = side == 0 ? Ji : Jj;
      for (int d = 0; d < 7; d++) {
        double add[7] = {0, 0, 0, 0, 0, 0, 0}, ep[7], em[7];
        add[d] = dlt; if (side == 0) edge_error(e, oplus(S[vi], add), S[vj], ep); else edge_error(e, S[vi], oplus(S[vj], add), ep);
        add[d] = -dlt; if (side == 0) edge_error(e, oplus(S[vi], add), S[vj], em); else edge_error(e, S[vi], oplus(S[vj], add), em);
        for (int k = 0; k < 7; k++) J[k * 7 + d] = scalar * (ep[k] - em[k]);
      }
    }
    auto acc = [&](int r, int c, const double* A, const double* B) {
      for (int p = 0; p < 7; p++) for (int q = 0; q < 7; q++) { double s2 = 0; for (int k = 0; k < 7; k++) s2 += A[k * 7 + p] * B[k * 7 + q]; Hdense[(7 * (size_t)r + p) * n + 7 * (size_t)c + q] += s2; }
    };
    if (a >= 0) { acc(a, a, Ji, Ji); for (int p = 0; p < 7; p++) for (int k = 0; k < 7; k++) bout[7 * (size_t)a + p] -= Ji[k * 7 + p] * er[k]; }
    if (b >= 0) { acc(b, b, Jj, Jj); for (int p = 0; p < 7; p++) for (int k = 0; k < 7; k++) bout[7 * (size_t)b + p] -= Jj[k * 7 + p] * er[k]; }
    if (a >= 0 && b >= 0) { acc(a, b, Ji, Jj); acc(b, a, Jj, Ji); }
  }
  return F;
}


// test hook: out = log(exp(u)) of g2o::Sim3 (sim3.h:72-140, 146-237)
void ora_sim3_exp_log(const double* u, double* out) { sim3_log(sim3_exp(u), out); }
// Converter::toSE3Quat (Converter.cc:40-50): f32 4x4 (row-major) -> Eigen::Quaterniond(R) [Eigen's trace method] -> SE3Quat(R, t) with
// normalizeRotation (se3quat.h:58-60, 280-285)
void ora_to_se3quat(const float* T, double* qt) {
  double m[3][3];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) m[i][j] = (double)T[4 * i + j];
  double q[4];
  double t = m[0][0] + m[1][1] + m[2][2];
  if (t > 0.0) {
    t = std::sqrt(t + 1.0); q[3] = 0.5 * t; t = 0.5 / t;
    q[0] = (m[2][1] - m[1][2]) * t; q[1] = (m[0][2] - m[2][0]) * t; q[2] = (m[1][0] - m[0][1]) * t;
  } else {
    int i = 0;
    if (m[1][1] > m[0][0]) i = 1;
    if (m[2][2] > m[i][i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(m[i][i] - m[j][j] - m[k][k] + 1.0);
    q[i] = 0.5 * t; t = 0.5 / t;
    q[3] = (m[k][j] - m[j][k]) * t; q[j] = (m[j][i] + m[i][j]) * t; q[k] = (m[k][i] + m[i][k]) * t;
  }
  Quat Q{q[0], q[1], q[2], q[3]};
  normalize_rotation(Q);
  qt[0] = Q.x; qt[1] = Q.y; qt[2] = Q.z; qt[3] = Q.w; qt[4] = (double)T[3]; qt[5] = (double)T[7]; qt[6] = (double)T[11];
}
// Converter::toCvMat(SE3Quat) (Converter.cc:52-56, 74-82): to_homogeneous_matrix (se3quat.h:271-277) rounded to f32
void ora_se3quat_to_cvmat(const double* qt, float* T) {
  const double x = qt[0], y = qt[1], z = qt[2], w = qt[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
  const double R[9] = {1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1 - (txx + tyy)};
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) T[4 * i + j] = (float)R[3 * i + j]; T[4 * i + 3] = (float)qt[4 + i]; }
  T[12] = 0.f; T[13] = 0.f; T[14] = 0.f; T[15] = 1.f;
}
// closed forms, for the pinning tests against oracle/_ref (tests/test_ref_g2o.py)
void ora_sim3_exp(const double* u, double* s8) {
  const Sim3 S = sim3_exp(u);
  s8[0] = S.r.x; s8[1] = S.r.y; s8[2] = S.r.z; s8[3] = S.r.w; s8[4] = S.t[0]; s8[5] = S.t[1]; s8[6] = S.t[2]; s8[7] = S.s;
}
void ora_se3_exp(const double* u, double* qt7) {
  const Pose T = se3_exp(u);
  qt7[0] = T.q.x; qt7[1] = T.q.y; qt7[2] = T.q.z; qt7[3] = T.q.w; qt7[4] = T.t[0]; qt7[5] = T.t[1]; qt7[6] = T.t[2];
}

}  // extern "C"
