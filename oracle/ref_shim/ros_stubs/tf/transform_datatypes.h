#pragma once
#include <tf/tf.h>
