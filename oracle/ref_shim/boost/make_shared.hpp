// look-alike of <boost/shared_ptr.hpp> for compiling reference sources (TEST INFRASTRUCTURE): boost::shared_ptr == std::shared_ptr
#pragma once
#include <memory>
namespace boost {
template <class T> using shared_ptr = std::shared_ptr<T>;
template <class T> using weak_ptr = std::weak_ptr<T>;
template <class T> using enable_shared_from_this = std::enable_shared_from_this<T>;
using std::const_pointer_cast;
using std::dynamic_pointer_cast;
using std::make_shared;
using std::static_pointer_cast;
}  // namespace boost
