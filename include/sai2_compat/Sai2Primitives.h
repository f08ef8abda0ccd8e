// Sai2Primitives.h — forwarding header: lets a program written against the reference's headers (src/Sai2Primitives.h, without the haptic classes: out of scope) compile with
// only the include path changed (-I <repo>/include/sai2_compat). Everything is declared in Sai2PrimitivesEigen.h.
#pragma once
#include "../Sai2PrimitivesEigen.h"
