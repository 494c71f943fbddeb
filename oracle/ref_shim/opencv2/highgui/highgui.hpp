// look-alike of <opencv2/highgui/highgui.hpp> (TEST INFRASTRUCTURE, see mini_cv.h)
#pragma once
#include "../mini_cv.h"
