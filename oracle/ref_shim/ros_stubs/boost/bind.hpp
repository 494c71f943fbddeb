// look-alike of <boost/bind.hpp> (TEST INFRASTRUCTURE): nothing of it is needed by the declarations the shim check parses
#pragma once
#include <functional>
