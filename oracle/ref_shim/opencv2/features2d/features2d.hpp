// look-alike of <opencv2/features2d/features2d.hpp> (TEST INFRASTRUCTURE, see mini_cv.h)
#pragma once
#include "../mini_cv.h"
