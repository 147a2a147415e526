// TEST INFRASTRUCTURE: what marks this library as the host (SIMT interpreter) build of the kernels.
// diamond_amd/native.py refuses to load a library that exports this symbol (the product has no CPU path).
extern "C" int dmd_simt_host_build(void) { return 1; }
