// look-alike of <cslam/Communicator.h> (TEST INFRASTRUCTURE): see MapGraph_lookalike.h
#pragma once
#include <cslam/MapGraph_lookalike.h>
