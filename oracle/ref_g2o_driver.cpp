// ref_g2o_driver.cpp — TEST INFRASTRUCTURE.  Feeds the reference's OWN g2o (compiled verbatim from
// /root/reference/cslam/thirdparty/g2o by oracle/Makefile.ref, against the look-alike Eigen in oracle/ref_shim/) with flat
// problems, building every graph the way cslam/src/Optimizer.cpp does (solver types, vertex ids, edge order, robust kernels,
// levels, optimize() calls).  The result, oracle/_ref/libg2o_ref.so, is what the oracle restatement oracle/ba_ref.cpp is
// pinned against (tests/test_ref_g2o.py).  Same flat layouts as the ora_* entry points of ba_ref.cpp.
// Nothing in the product links or loads this.
#include <cstdint>
#include <cstring>
#include <vector>

#include "core/block_solver.h"
#include "core/optimization_algorithm_levenberg.h"
#include "core/robust_kernel_impl.h"
#include "core/sparse_optimizer.h"
#include "core/batch_stats.h"
#include "core/jacobian_workspace.h"
#include "solvers/linear_solver_dense.h"
#include "solvers/linear_solver_eigen.h"
#include "types/types_seven_dof_expmap.h"
#include "types/types_six_dof_expmap.h"

namespace {
g2o::SE3Quat se3_from(const double* v) {   // rows: qx qy qz qw tx ty tz; the SE3Quat(q, t) ctor normalises like Converter::toSE3Quat's result
  return g2o::SE3Quat(Eigen::Quaterniond(v[3], v[0], v[1], v[2]), Eigen::Vector3d(v[4], v[5], v[6]));
}
void se3_to(const g2o::SE3Quat& T, double* v) {
  v[0] = T.rotation().x(); v[1] = T.rotation().y(); v[2] = T.rotation().z(); v[3] = T.rotation().w();
  v[4] = T.translation()[0]; v[5] = T.translation()[1]; v[6] = T.translation()[2];
}
g2o::Sim3 sim3_from(const double* p) { return g2o::Sim3(Eigen::Quaterniond(p[3], p[0], p[1], p[2]), Eigen::Vector3d(p[4], p[5], p[6]), p[7]); }
void sim3_to(const g2o::Sim3& S, double* p) {
  p[0] = S.rotation().x(); p[1] = S.rotation().y(); p[2] = S.rotation().z(); p[3] = S.rotation().w();
  p[4] = S.translation()[0]; p[5] = S.translation()[1]; p[6] = S.translation()[2]; p[7] = S.scale();
}
}  // namespace

extern "C" {

struct ref_ba_stats {
  int32_t iters_done, lm_trials, n_hist;
  double chi2_final, lambda_final;
  double chi2_hist[64];        // activeRobustChi2 after every iteration (G2OBatchStatistics::chi2)
  int32_t trials_hist[64];     // OptimizationAlgorithmLevenberg::levenbergIterations() of every iteration
};

// Optimizer::MapFusionGBA (Optimizer.cpp:646-859) / LocalBundleAdjustmentClient (:349-644) / BundleAdjustmentClient (:40-212) on a flat
// problem: BlockSolver_6_3 + LinearSolverEigen (dense_solver != 0: LinearSolverDense) + Levenberg; camera vertices first (id = index),
// then marginalised point vertices (id = n_cam + index); EdgeSE3ProjectXYZ with vertex(0) = point, vertex(1) = camera, information
// I2 * invSigma2, Huber(delta) when delta > 0, level from e_level; initializeOptimization(0); optimize(max_iters).
int ref_ba_optimize(int n_cam, int n_pt, int n_edge, double* cam_qt, const uint8_t* cam_fixed, const double* cam_K, double* pt_xyz,
                    const int32_t* e_cam, const int32_t* e_pt, const double* e_obs, const double* e_info, const uint8_t* e_level,
                    double huber_delta, int max_iters, int dense_solver, double lambda_init, bool* stop_flag, double* chi2_per_edge,
                    uint8_t* depth_pos, ref_ba_stats* stats) {
  g2o::SparseOptimizer optimizer;
  g2o::BlockSolver_6_3::LinearSolverType* linearSolver;
  if (dense_solver) linearSolver = new g2o::LinearSolverDense<g2o::BlockSolver_6_3::PoseMatrixType>();
  else linearSolver = new g2o::LinearSolverEigen<g2o::BlockSolver_6_3::PoseMatrixType>();
  g2o::BlockSolver_6_3* solver_ptr = new g2o::BlockSolver_6_3(linearSolver);
  g2o::OptimizationAlgorithmLevenberg* solver = new g2o::OptimizationAlgorithmLevenberg(solver_ptr);
  if (lambda_init > 0) solver->setUserLambdaInit(lambda_init);
  optimizer.setAlgorithm(solver);
  if (stop_flag) optimizer.setForceStopFlag(stop_flag);
  for (int c = 0; c < n_cam; c++) {
    g2o::VertexSE3Expmap* vSE3 = new g2o::VertexSE3Expmap();
    vSE3->setEstimate(se3_from(cam_qt + 7 * (size_t)c));
    vSE3->setId(c);
    vSE3->setFixed(cam_fixed[c] != 0);
    optimizer.addVertex(vSE3);
  }
  std::vector<char> used(n_pt, 0);
  for (int e = 0; e < n_edge; e++) used[e_pt[e]] = 1;
  for (int p = 0; p < n_pt; p++) {
    if (!used[p]) continue;
    g2o::VertexSBAPointXYZ* vPoint = new g2o::VertexSBAPointXYZ();
    vPoint->setEstimate(Eigen::Vector3d(pt_xyz[3 * (size_t)p], pt_xyz[3 * (size_t)p + 1], pt_xyz[3 * (size_t)p + 2]));
    vPoint->setId(n_cam + p);
    vPoint->setMarginalized(true);
    optimizer.addVertex(vPoint);
  }
  std::vector<g2o::EdgeSE3ProjectXYZ*> edges(n_edge);
  for (int k = 0; k < n_edge; k++) {
    g2o::EdgeSE3ProjectXYZ* e = new g2o::EdgeSE3ProjectXYZ();
    e->setVertex(0, dynamic_cast<g2o::OptimizableGraph::Vertex*>(optimizer.vertex(n_cam + e_pt[k])));
    e->setVertex(1, dynamic_cast<g2o::OptimizableGraph::Vertex*>(optimizer.vertex(e_cam[k])));
    Eigen::Matrix<double, 2, 1> obs(e_obs[2 * (size_t)k], e_obs[2 * (size_t)k + 1]);
    e->setMeasurement(obs);
    e->setInformation(Eigen::Matrix2d::Identity() * e_info[k]);
    if (huber_delta > 0) {
      g2o::RobustKernelHuber* rk = new g2o::RobustKernelHuber;
      e->setRobustKernel(rk);
      rk->setDelta(huber_delta);
    }
    const double* K = cam_K + 4 * (size_t)e_cam[k];
    e->fx = K[0]; e->fy = K[1]; e->cx = K[2]; e->cy = K[3];
    if (e_level) e->setLevel(e_level[k]);
    optimizer.addEdge(e);
    edges[k] = e;
  }
  optimizer.setComputeBatchStatistics(true);
  optimizer.initializeOptimization(0);
  const int its = optimizer.optimize(max_iters);
  for (int c = 0; c < n_cam; c++) se3_to(static_cast<g2o::VertexSE3Expmap*>(optimizer.vertex(c))->estimate(), cam_qt + 7 * (size_t)c);
  for (int p = 0; p < n_pt; p++) {
    if (!used[p]) continue;
    const Eigen::Vector3d x = static_cast<g2o::VertexSBAPointXYZ*>(optimizer.vertex(n_cam + p))->estimate();
    pt_xyz[3 * (size_t)p] = x[0]; pt_xyz[3 * (size_t)p + 1] = x[1]; pt_xyz[3 * (size_t)p + 2] = x[2];
  }
  for (int k = 0; k < n_edge; k++) {
    if (chi2_per_edge && edges[k]->level() == 0) chi2_per_edge[k] = edges[k]->chi2();   // what Optimizer.cpp reads after optimize() (:574-602)
    if (depth_pos) depth_pos[k] = edges[k]->isDepthPositive() ? 1 : 0;
  }
  if (stats) {
    std::memset(stats, 0, sizeof(*stats));
    stats->iters_done = its;
    const g2o::BatchStatisticsContainer& bs = optimizer.batchStatistics();
    for (size_t i = 0; i < bs.size() && (int)i < its && i < 64; i++) {
      stats->chi2_hist[i] = bs[i].chi2; stats->trials_hist[i] = bs[i].levenbergIterations; stats->lm_trials += bs[i].levenbergIterations;
      stats->n_hist = (int)i + 1;
    }
    stats->chi2_final = optimizer.activeRobustChi2();
    stats->lambda_final = solver->currentLambda();
  }
  return its;
}

// Optimizer::PoseOptimizationClient (Optimizer.cpp:215-347)
int ref_pose_optimize(double* cam_qt, int n, const double* Xw, const double* obs, const double* info, const double* Kc, uint8_t* outlier) {
  g2o::SparseOptimizer optimizer;
  g2o::BlockSolver_6_3::LinearSolverType* linearSolver = new g2o::LinearSolverDense<g2o::BlockSolver_6_3::PoseMatrixType>();
  g2o::BlockSolver_6_3* solver_ptr = new g2o::BlockSolver_6_3(linearSolver);
  g2o::OptimizationAlgorithmLevenberg* solver = new g2o::OptimizationAlgorithmLevenberg(solver_ptr);
  optimizer.setAlgorithm(solver);
  int nInitialCorrespondences = 0;
  const g2o::SE3Quat Tcw = se3_from(cam_qt);
  g2o::VertexSE3Expmap* vSE3 = new g2o::VertexSE3Expmap();
  vSE3->setEstimate(Tcw);
  vSE3->setId(0);
  vSE3->setFixed(false);
  optimizer.addVertex(vSE3);
  std::vector<g2o::EdgeSE3ProjectXYZOnlyPose*> vpEdgesMono;
  const float deltaMono = sqrt(5.991);
  for (int i = 0; i < n; i++) {
    nInitialCorrespondences++;
    outlier[i] = 0;
    Eigen::Matrix<double, 2, 1> o(obs[2 * (size_t)i], obs[2 * (size_t)i + 1]);
    g2o::EdgeSE3ProjectXYZOnlyPose* e = new g2o::EdgeSE3ProjectXYZOnlyPose();
    e->setVertex(0, dynamic_cast<g2o::OptimizableGraph::Vertex*>(optimizer.vertex(0)));
    e->setMeasurement(o);
    e->setInformation(Eigen::Matrix2d::Identity() * info[i]);
    g2o::RobustKernelHuber* rk = new g2o::RobustKernelHuber;
    e->setRobustKernel(rk);
    rk->setDelta(deltaMono);
    e->fx = Kc[0]; e->fy = Kc[1]; e->cx = Kc[2]; e->cy = Kc[3];
    e->Xw[0] = Xw[3 * (size_t)i]; e->Xw[1] = Xw[3 * (size_t)i + 1]; e->Xw[2] = Xw[3 * (size_t)i + 2];
    optimizer.addEdge(e);
    vpEdgesMono.push_back(e);
  }
  if (nInitialCorrespondences < 3) return 0;
  const float chi2Mono[4] = {5.991, 5.991, 5.991, 5.991};
  const int its[4] = {10, 10, 10, 10};
  int nBad = 0;
  for (size_t it = 0; it < 4; it++) {
    vSE3->setEstimate(Tcw);
    optimizer.initializeOptimization(0);
    optimizer.optimize(its[it]);
    nBad = 0;
    for (size_t i = 0, iend = vpEdgesMono.size(); i < iend; i++) {
      g2o::EdgeSE3ProjectXYZOnlyPose* e = vpEdgesMono[i];
      if (outlier[i]) e->computeError();
      const float chi2 = e->chi2();
      if (chi2 > chi2Mono[it]) { outlier[i] = 1; e->setLevel(1); nBad++; }
      else { outlier[i] = 0; e->setLevel(0); }
      if (it == 2) e->setRobustKernel(0);
    }
    if (optimizer.edges().size() < 10) break;
  }
  se3_to(static_cast<g2o::VertexSE3Expmap*>(optimizer.vertex(0))->estimate(), cam_qt);
  return nInitialCorrespondences - nBad;
}

// Optimizer::OptimizeSim3 (Optimizer.cpp:861-1056) on the pairs that survive its null / bad / i2 < 0 filter
int ref_sim3_optimize(double* sim3, int n, const double* P1c, const double* P2c, const double* obs1, const double* obs2, const double* info1,
                      const double* info2, const double* K1, const double* K2, double th2_d, int fix_scale, uint8_t* inlier) {
  const float th2 = (float)th2_d;
  g2o::SparseOptimizer optimizer;
  g2o::BlockSolverX::LinearSolverType* linearSolver = new g2o::LinearSolverDense<g2o::BlockSolverX::PoseMatrixType>();
  g2o::BlockSolverX* solver_ptr = new g2o::BlockSolverX(linearSolver);
  g2o::OptimizationAlgorithmLevenberg* solver = new g2o::OptimizationAlgorithmLevenberg(solver_ptr);
  optimizer.setAlgorithm(solver);
  g2o::VertexSim3Expmap* vSim3 = new g2o::VertexSim3Expmap();
  vSim3->_fix_scale = fix_scale != 0;
  vSim3->setEstimate(sim3_from(sim3));
  vSim3->setId(0);
  vSim3->setFixed(false);
  vSim3->_principle_point1[0] = K1[2]; vSim3->_principle_point1[1] = K1[3];
  vSim3->_focal_length1[0] = K1[0]; vSim3->_focal_length1[1] = K1[1];
  vSim3->_principle_point2[0] = K2[2]; vSim3->_principle_point2[1] = K2[3];
  vSim3->_focal_length2[0] = K2[0]; vSim3->_focal_length2[1] = K2[1];
  optimizer.addVertex(vSim3);
  std::vector<g2o::EdgeSim3ProjectXYZ*> vpEdges12;
  std::vector<g2o::EdgeInverseSim3ProjectXYZ*> vpEdges21;
  const float deltaHuber = sqrt(th2);
  int nCorrespondences = 0;
  for (int i = 0; i < n; i++) {
    const int id1 = 2 * i + 1, id2 = 2 * (i + 1);
    g2o::VertexSBAPointXYZ* vPoint1 = new g2o::VertexSBAPointXYZ();
    vPoint1->setEstimate(Eigen::Vector3d(P1c[3 * (size_t)i], P1c[3 * (size_t)i + 1], P1c[3 * (size_t)i + 2]));
    vPoint1->setId(id1);
    vPoint1->setFixed(true);
    optimizer.addVertex(vPoint1);
    g2o::VertexSBAPointXYZ* vPoint2 = new g2o::VertexSBAPointXYZ();
    vPoint2->setEstimate(Eigen::Vector3d(P2c[3 * (size_t)i], P2c[3 * (size_t)i + 1], P2c[3 * (size_t)i + 2]));
    vPoint2->setId(id2);
    vPoint2->setFixed(true);
    optimizer.addVertex(vPoint2);
    nCorrespondences++;
    Eigen::Matrix<double, 2, 1> o1(obs1[2 * (size_t)i], obs1[2 * (size_t)i + 1]);
    g2o::EdgeSim3ProjectXYZ* e12 = new g2o::EdgeSim3ProjectXYZ();
    e12->setVertex(0, dynamic_cast<g2o::OptimizableGraph::Vertex*>(optimizer.vertex(id2)));
    e12->setVertex(1, dynamic_cast<g2o::OptimizableGraph::Vertex*>(optimizer.vertex(0)));
    e12->setMeasurement(o1);
    e12->setInformation(Eigen::Matrix2d::Identity() * info1[i]);
    g2o::RobustKernelHuber* rk1 = new g2o::RobustKernelHuber;
    e12->setRobustKernel(rk1);
    rk1->setDelta(deltaHuber);
    optimizer.addEdge(e12);
    Eigen::Matrix<double, 2, 1> o2(obs2[2 * (size_t)i], obs2[2 * (size_t)i + 1]);
    g2o::EdgeInverseSim3ProjectXYZ* e21 = new g2o::EdgeInverseSim3ProjectXYZ();
    e21->setVertex(0, dynamic_cast<g2o::OptimizableGraph::Vertex*>(optimizer.vertex(id1)));
    e21->setVertex(1, dynamic_cast<g2o::OptimizableGraph::Vertex*>(optimizer.vertex(0)));
    e21->setMeasurement(o2);
    e21->setInformation(Eigen::Matrix2d::Identity() * info2[i]);
    g2o::RobustKernelHuber* rk2 = new g2o::RobustKernelHuber;
    e21->setRobustKernel(rk2);
    rk2->setDelta(deltaHuber);
    optimizer.addEdge(e21);
    vpEdges12.push_back(e12);
    vpEdges21.push_back(e21);
    inlier[i] = 1;
  }
  optimizer.initializeOptimization();
  optimizer.optimize(5);
  int nBad = 0;
  for (size_t i = 0; i < vpEdges12.size(); i++) {
    g2o::EdgeSim3ProjectXYZ* e12 = vpEdges12[i];
    g2o::EdgeInverseSim3ProjectXYZ* e21 = vpEdges21[i];
    if (!e12 || !e21) continue;
    if (e12->chi2() > th2 || e21->chi2() > th2) {
      inlier[i] = 0;
      optimizer.removeEdge(e12);
      optimizer.removeEdge(e21);
      vpEdges12[i] = static_cast<g2o::EdgeSim3ProjectXYZ*>(NULL);
      vpEdges21[i] = static_cast<g2o::EdgeInverseSim3ProjectXYZ*>(NULL);
      nBad++;
    }
  }
  const int nMoreIterations = nBad > 0 ? 10 : 5;
  if (nCorrespondences - nBad < 10) return 0;
  optimizer.initializeOptimization();
  optimizer.optimize(nMoreIterations);
  int nIn = 0;
  for (size_t i = 0; i < vpEdges12.size(); i++) {
    g2o::EdgeSim3ProjectXYZ* e12 = vpEdges12[i];
    g2o::EdgeInverseSim3ProjectXYZ* e21 = vpEdges21[i];
    if (!e12 || !e21) continue;
    if (e12->chi2() > th2 || e21->chi2() > th2) inlier[i] = 0;
    else nIn++;
  }
  sim3_to(static_cast<g2o::VertexSim3Expmap*>(optimizer.vertex(0))->estimate(), sim3);
  return nIn;
}

struct ref_pg_stats { int32_t iters_done, lm_trials; double chi2_initial, chi2_final, lambda_final; };

// g2o part of Optimizer::OptimizeEssentialGraphLoopClosure / MapFusion (Optimizer.cpp:1058-1266, 1333-1500): BlockSolver_7_3 +
// LinearSolverEigen + Levenberg with setUserLambdaInit(1e-16), VertexSim3Expmap (not marginalised), EdgeSim3 with information I7,
// vertex(0) = i, vertex(1) = j, initializeOptimization(); optimize(20).
int ref_pose_graph_optimize(int n_vert, double* sim3, const uint8_t* fixed, int fix_scale, int n_edge, const int32_t* e_i, const int32_t* e_j,
                            const double* meas, int max_iters, double lambda_init, ref_pg_stats* stats) {
  g2o::SparseOptimizer optimizer;
  optimizer.setVerbose(false);
  g2o::BlockSolver_7_3::LinearSolverType* linearSolver = new g2o::LinearSolverEigen<g2o::BlockSolver_7_3::PoseMatrixType>();
  g2o::BlockSolver_7_3* solver_ptr = new g2o::BlockSolver_7_3(linearSolver);
  g2o::OptimizationAlgorithmLevenberg* solver = new g2o::OptimizationAlgorithmLevenberg(solver_ptr);
  if (lambda_init > 0) solver->setUserLambdaInit(lambda_init);
  optimizer.setAlgorithm(solver);
  for (int v = 0; v < n_vert; v++) {
    g2o::VertexSim3Expmap* VSim3 = new g2o::VertexSim3Expmap();
    VSim3->setEstimate(sim3_from(sim3 + 8 * (size_t)v));
    if (fixed[v]) VSim3->setFixed(true);
    VSim3->setId(v);
    VSim3->setMarginalized(false);
    VSim3->_fix_scale = fix_scale != 0;
    optimizer.addVertex(VSim3);
  }
  const Eigen::Matrix<double, 7, 7> matLambda = Eigen::Matrix<double, 7, 7>::Identity();
  for (int k = 0; k < n_edge; k++) {
    g2o::EdgeSim3* e = new g2o::EdgeSim3();
    e->setVertex(1, dynamic_cast<g2o::OptimizableGraph::Vertex*>(optimizer.vertex(e_j[k])));
    e->setVertex(0, dynamic_cast<g2o::OptimizableGraph::Vertex*>(optimizer.vertex(e_i[k])));
    e->setMeasurement(sim3_from(meas + 8 * (size_t)k));
    e->information() = matLambda;
    optimizer.addEdge(e);
  }
  optimizer.setComputeBatchStatistics(true);
  optimizer.initializeOptimization();
  optimizer.computeActiveErrors();
  const double chi0 = optimizer.activeRobustChi2();
  const int its = optimizer.optimize(max_iters);
  for (int v = 0; v < n_vert; v++) sim3_to(static_cast<g2o::VertexSim3Expmap*>(optimizer.vertex(v))->estimate(), sim3 + 8 * (size_t)v);
  if (stats) {
    stats->iters_done = its; stats->lm_trials = 0;
    const g2o::BatchStatisticsContainer& bs = optimizer.batchStatistics();
    for (size_t i = 0; i < bs.size() && (int)i < its; i++) stats->lm_trials += bs[i].levenbergIterations;
    stats->chi2_initial = chi0; stats->chi2_final = optimizer.activeRobustChi2(); stats->lambda_final = solver->currentLambda();
  }
  return its;
}

// SE3Quat / Sim3 exponential and logarithm maps (se3quat.h:175-257, sim3.h:72-237) for direct formula checks
void ref_se3_exp(const double* u6, double* qt7) { Eigen::Matrix<double, 6, 1> u; for (int i = 0; i < 6; i++) u[i] = u6[i]; se3_to(g2o::SE3Quat::exp(u), qt7); }
void ref_se3_log(const double* qt7, double* u6) { const Eigen::Matrix<double, 6, 1> u = se3_from(qt7).log(); for (int i = 0; i < 6; i++) u6[i] = u[i]; }
void ref_sim3_exp(const double* u7, double* s8) { Eigen::Matrix<double, 7, 1> u; for (int i = 0; i < 7; i++) u[i] = u7[i]; sim3_to(g2o::Sim3(u), s8); }
void ref_sim3_log(const double* s8, double* u7) { const Eigen::Matrix<double, 7, 1> u = sim3_from(s8).log(); for (int i = 0; i < 7; i++) u7[i] = u[i]; }
// RobustKernelHuber::robustify (robust_kernel_impl.cpp:78-90)
void ref_huber(double delta, double e2, double* rho3) {
  g2o::RobustKernelHuber rk; rk.setDelta(delta);
  Eigen::Vector3d rho; rk.robustify(e2, rho);
  rho3[0] = rho[0]; rho3[1] = rho[1]; rho3[2] = rho[2];
}
// EdgeSE3ProjectXYZ: error and analytic Jacobians at a state (types_six_dof_expmap.cpp:103-147, .h:90-95)
void ref_edge_se3_project(const double* cam_qt, const double* K, const double* X, const double* obs, double* err2, double* Jpt_2x3, double* Jcam_2x6) {
  g2o::VertexSE3Expmap vc; vc.setEstimate(se3_from(cam_qt)); vc.setId(1);
  g2o::VertexSBAPointXYZ vp; vp.setEstimate(Eigen::Vector3d(X[0], X[1], X[2])); vp.setId(0);
  g2o::EdgeSE3ProjectXYZ e;
  e.setVertex(0, &vp); e.setVertex(1, &vc);
  e.setMeasurement(Eigen::Matrix<double, 2, 1>(obs[0], obs[1]));
  e.fx = K[0]; e.fy = K[1]; e.cx = K[2]; e.cy = K[3];
  e.computeError();
  g2o::JacobianWorkspace ws;   // the edge's Jacobians are maps into workspace memory (base_binary_edge.hpp:122-127)
  ws.updateSize(&e);
  ws.allocate();
  static_cast<g2o::OptimizableGraph::Edge&>(e).linearizeOplus(ws);   // the workspace overload is hidden by the edge's own linearizeOplus()
  err2[0] = e.error()[0]; err2[1] = e.error()[1];
  for (int r = 0; r < 2; r++) for (int c = 0; c < 3; c++) Jpt_2x3[3 * r + c] = e.jacobianOplusXi()(r, c);
  for (int r = 0; r < 2; r++) for (int c = 0; c < 6; c++) Jcam_2x6[6 * r + c] = e.jacobianOplusXj()(r, c);
}

}  // extern "C"
