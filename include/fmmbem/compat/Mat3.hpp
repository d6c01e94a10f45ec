// Forwarding header (see FMM_plan.hpp in this directory): the adapter's Vec<N,T> / Mat3<T> under the reference's file name.
#pragma once
#include "../Vec.hpp"
