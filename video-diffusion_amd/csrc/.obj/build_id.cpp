extern "C" const char* vd_source_sha(void) { return "8aa51d74ba317c45"; }
