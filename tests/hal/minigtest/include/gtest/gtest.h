// TEST INFRASTRUCTURE: some unit tests of the reference include googletest under this path
#include "../../gtest/gtest.h"
