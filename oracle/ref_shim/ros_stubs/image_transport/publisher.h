#pragma once
#include <image_transport/image_transport.h>
