// asan_stubs.cpp -- only linked into the host-side sanitizer build (make asan; tests/test_sanitizers.py).
// That build compiles the host pass alone, so the fat binary the kernels' host stubs would register does not exist:
// its symbols are defined here as dummies (generated list) and the registration hooks are no-ops.  Nothing can be
// launched from this library; it exists to run the C-ABI's host code under AddressSanitizer / UBSan on a CPU box.
#include "asan_fatbin_syms.inc"

extern "C" {
void** __hipRegisterFatBinary(const void*) { static void* handle = nullptr; return &handle; }
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, char*, int, unsigned long, int, int) {}
}
