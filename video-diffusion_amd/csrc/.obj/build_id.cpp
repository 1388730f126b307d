extern "C" const char* vd_source_sha(void) { return "5c1af29f18c1c358"; }
