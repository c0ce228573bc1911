// gmx_internal.h — glue shared by the two translation units of libgmx.so.
#pragma once
#include <string>

#include "../../include/gmx.h"
#include "gmx_index.h"

void gmx_set_error(const std::string &msg);
const gmx::HostIndex &gmx_index_host(const gmx_index *ix);
