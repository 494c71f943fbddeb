#pragma once
#include <tf/tf.h>
namespace tf { struct TransformBroadcaster {}; }
