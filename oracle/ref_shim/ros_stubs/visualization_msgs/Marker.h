// look-alike of <visualization_msgs/Marker.h> (TEST INFRASTRUCTURE)
#pragma once
#include <vector>
#include <string>
#include <std_msgs/Header.h>
#include <std_msgs/ColorRGBA.h>
#include <geometry_msgs/Point.h>
namespace visualization_msgs {
struct Marker {
  enum { ARROW = 0, CUBE = 1, SPHERE = 2, CYLINDER = 3, LINE_STRIP = 4, LINE_LIST = 5, CUBE_LIST = 6, SPHERE_LIST = 7, POINTS = 8, ADD = 0, MODIFY = 0, DELETE = 2 };
  std_msgs::Header header; std::string ns; int id = 0, type = 0, action = 0; geometry_msgs::Pose pose; geometry_msgs::Vector3 scale; std_msgs::ColorRGBA color;
  std::vector<geometry_msgs::Point> points; std::vector<std_msgs::ColorRGBA> colors;
};
}
