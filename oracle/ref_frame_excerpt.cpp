// ref_frame_excerpt.cpp — TEST INFRASTRUCTURE.  The grid functions of the reference's Frame and KeyFrame — Frame::AssignFeaturesToGrid
// (cslam/src/Frame.cpp:103-118), Frame::GetFeaturesInArea + PosInGrid (:200-265), KeyFrame::GetFeaturesInArea + IsInImage (KeyFrame.cpp:1162-1206)
// — extracted at build time by oracle/Makefile.ref into oracle/_ref/gen/*.inc and compiled as members of the look-alike classes, whose data
// members carry the reference's names.  They define the candidate ORDER of every window search.
// Also Frame::UpdatePoseMatrices + Frame::isInFrustum (Frame.cpp:131-198) and MapPoint::ComputeDistinctiveDescriptors (MapPoint.cpp:929-994), which calls the reference's ORBmatcher::DescriptorDistance (linked from ORBmatcher.cpp).
#include <climits>
#include <cmath>
#include <cslam/Frame.h>
#include <cslam/KeyFrame.h>
#include <cslam/MapPoint.h>
#include <cslam/ORBmatcher.h>
using namespace std;
namespace cslam {
float Frame::mfGridElementWidthInv, Frame::mfGridElementHeightInv, Frame::mnMinX, Frame::mnMaxX, Frame::mnMinY, Frame::mnMaxY;
#include "Frame_103_118.inc"
typedef boost::shared_ptr<MapPoint> mpptr;
#include "Frame_131_198.inc"      // UpdatePoseMatrices, isInFrustum
#include "Frame_200_265.inc"
#include "KeyFrame_1162_1206.inc"
typedef boost::shared_ptr<KeyFrame> kfptr;
#include "MapPoint_929_994.inc"
}  // namespace cslam
