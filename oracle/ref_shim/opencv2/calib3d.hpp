// look-alike of <opencv2/calib3d.hpp> (TEST INFRASTRUCTURE, see mini_cv.h)
#pragma once
#include "mini_cv.h"
