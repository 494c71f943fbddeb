#pragma once
#include "point_cloud.h"
