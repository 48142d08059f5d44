// Additional kernel modes built from the same building blocks (k -> k_f, dk_f accumulation,
// dk_f -> dk).  Filled in below.
#pragma once
#include "ffc_body.h"
