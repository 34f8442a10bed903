#include "cereal/cereal.hpp"
