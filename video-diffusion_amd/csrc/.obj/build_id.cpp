extern "C" const char* vd_source_sha(void) { return "191e07a2e728d98a"; }
