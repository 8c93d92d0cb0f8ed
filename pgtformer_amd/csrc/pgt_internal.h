// Internal umbrella: public C-ABI + helpers shared by the translation units of libpgt_hip.so.
#pragma once
#include "../../include/pgt_hip.h"
