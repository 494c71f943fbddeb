#pragma once
#include <visualization_msgs/Marker.h>
namespace visualization_msgs { struct MarkerArray { std::vector<Marker> markers; }; }
