// look-alike (TEST INFRASTRUCTURE): nothing of PCL is used by the compiled reference sources
#pragma once
namespace pcl { struct PointXYZ { float x, y, z; }; struct PointXYZRGB { float x, y, z; unsigned rgb; }; template <class T> struct PointCloud {}; }
