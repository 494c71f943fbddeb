// look-alike of <opencv2/imgproc/imgproc.hpp> (TEST INFRASTRUCTURE, see mini_cv.h)
#pragma once
#include "../mini_cv.h"
