// shape_inference lives in the functional model of op_kernel.h (tests/tf_mock)
#pragma once
#include "tensorflow/core/framework/op_kernel.h"
