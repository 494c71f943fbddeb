// look-alike of <tf/tf.h> (TEST INFRASTRUCTURE)
#pragma once
namespace tf { struct Vector3 {}; struct Quaternion {}; struct Matrix3x3 {}; struct Transform {}; struct StampedTransform : Transform {}; }
