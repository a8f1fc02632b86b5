// operator apply, 3_10 (2 top + 4 side streams per direction pair): see tsx_spmv_impl.hpp
#define TSX_SPMV_NTOP 2
#define TSX_SPMV_TAG 310
#include "tsx_spmv_impl.hpp"

TSX_CODE_PROBE(spmv310)  // tsx_host.hpp: this unit's code object as it sits in device memory (diagnostics)
