// <opencv2/opencv.hpp> for the check against the reference's REAL class headers (TEST INFRASTRUCTURE): the look-alike of oracle/ref_shim plus the legacy
// C-API type PnPSolver.h names in its declarations
#pragma once
#include_next <opencv2/opencv.hpp>
struct CvMat { int rows = 0, cols = 0; union { double* db; float* fl; unsigned char* ptr; } data; };
