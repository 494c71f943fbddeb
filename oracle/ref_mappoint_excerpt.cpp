// ref_mappoint_excerpt.cpp — TEST INFRASTRUCTURE.  MapPoint::UpdateNormalAndDepth of the reference (cslam/src/MapPoint.cpp:779-823), extracted
// at build time by oracle/Makefile.ref into oracle/_ref/gen/MapPoint_779_823.inc and compiled as a member of the look-alike MapPoint
// (oracle/ref_shim/cslam_lookalike/cslam/MapGraph_lookalike.h), whose data members carry the reference's names.  Nothing of the reference is
// stored in this repository.
#include <cslam/MapPoint.h>
#include <cslam/KeyFrame.h>
#include <cslam/Frame.h>
using namespace std;
namespace cslam {
typedef boost::shared_ptr<KeyFrame> kfptr;
typedef boost::shared_ptr<Frame> frameptr;
#include "MapPoint_779_823.inc"
#include "MapPoint_837_869.inc"   // PredictScale(dist, kfptr) and PredictScale(dist, frameptr)
}  // namespace cslam
