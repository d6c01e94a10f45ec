// Forwarding header: lets a translation unit written against the reference's include paths -- `#include "FMM_plan.hpp"` -- pick
// up the MI355X adapter instead ("Building the reference's own drivers").  Put this directory FIRST on the include path.
#pragma once
#include "../FMM_plan.hpp"
