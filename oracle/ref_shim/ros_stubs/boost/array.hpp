// look-alike of <boost/array.hpp> (TEST INFRASTRUCTURE)
#pragma once
#include <array>
namespace boost { template <class T, std::size_t N> using array = std::array<T, N>; }
