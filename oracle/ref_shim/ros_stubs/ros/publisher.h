#pragma once
#include <ros/ros.h>
