// helper_modules/Sai2PrimitivesCommonDefinitions.h — forwarding header: lets a program written against the reference's headers (src/helper_modules/Sai2PrimitivesCommonDefinitions.h) compile with
// only the include path changed (-I <repo>/include/sai2_compat). Everything is declared in Sai2PrimitivesEigen.h.
#pragma once
#include "../../Sai2PrimitivesEigen.h"
