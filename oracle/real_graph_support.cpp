// real_graph_support.cpp — TEST INFRASTRUCTURE for shim/liboptimizer_hip_shim_real.so: what the reference's REAL KeyFrame.cpp / MapPoint.cpp / Map.cpp /
// Frame.cpp (compiled as they are by shim/Makefile) still need at link time and that has no place in a test: the three Communicator entry points through which a
// map object queues itself for publication (Communicator.cpp needs ROS), KeyFrameDatabase::erase (Database.cpp needs the vocabulary), and cv::undistortPoints
// (OpenCV is not in the image).  The Communicator stubs count their calls so that a test can see that the shim's write-back reached the reference's
// SendMe() path exactly as Optimizer.cpp's does.
#include <atomic>
#include <stdexcept>

#include <cslam/Communicator.h>
#include <cslam/Database.h>

namespace {
std::atomic<long> g_pass_kf{0}, g_pass_mp{0}, g_del_mp{0}, g_db_erase{0};
}

namespace cslam {
void Communicator::PassKftoComm(kfptr) { g_pass_kf++; }
void Communicator::PassMptoComm(mpptr) { g_pass_mp++; }
void Communicator::DeleteMpFromBuffer(mpptr) { g_del_mp++; }
void KeyFrameDatabase::erase(kfptr) { g_db_erase++; }
}  // namespace cslam

namespace cv {
void undistortPoints(const Mat&, Mat&, const Mat&, const Mat&, const Mat&, const Mat&) {
  throw std::runtime_error("cv::undistortPoints: not part of the real-class harness (Frame construction from an image is not exercised)");
}
}  // namespace cv

extern "C" void mapg_comm_counters(long out[4]) { out[0] = g_pass_kf; out[1] = g_pass_mp; out[2] = g_del_mp; out[3] = g_db_erase; }
extern "C" int mapg_uses_real_classes(void) { return 1; }
