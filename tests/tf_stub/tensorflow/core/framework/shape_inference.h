#pragma once
#include "tensorflow/core/framework/op_kernel.h"
