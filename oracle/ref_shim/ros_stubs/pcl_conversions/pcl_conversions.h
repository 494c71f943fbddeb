#pragma once
#include <pcl/point_cloud.h>
