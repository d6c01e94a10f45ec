// Forwarding header (see FMM_plan.hpp in this directory): the adapter's Vec<N,T> under the reference's file name -- plus what
// the reference's own files take from its Vec.hpp by transitive inclusion (examples/BEM/Triangulation.hpp:196 calls an
// unqualified isnan, EvalP2P.hpp:88 likewise; include/Vec.hpp reaches <cmath> and friends through Boost).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <iostream>
#include <iterator>
#include <numeric>

#include "../Vec.hpp"
using std::isnan;
