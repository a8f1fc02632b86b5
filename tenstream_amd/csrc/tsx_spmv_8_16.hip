// operator apply, 8_16 (8 top + 4 side streams per direction pair): see tsx_spmv_impl.hpp
#define TSX_SPMV_NTOP 8
#define TSX_SPMV_TAG 816
#include "tsx_spmv_impl.hpp"

TSX_CODE_PROBE(spmv816)  // tsx_host.hpp: this unit's code object as it sits in device memory (diagnostics)
