// Sai2Model.h — forwarding header for programs that include sai2-model's "Sai2Model.h" on a machine without
// sai2-model: Sai2PrimitivesEigen.h declares the subset of Sai2Model::Sai2Model the control path touches (state, dof,
// updateModel, frame poses, setTRobotBase). With the real sai2-model installed, leave this directory's Sai2Model.h off
// the include path and define SAI2B_EXTERNAL_SAI2_MODEL (see Sai2PrimitivesEigen.h).
#pragma once
#include "../Sai2PrimitivesEigen.h"
