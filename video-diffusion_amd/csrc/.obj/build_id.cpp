extern "C" const char* vd_source_sha(void) { return "8f7719b008d10f7a"; }
