extern "C" const char* vd_source_sha(void) { return "da92a0ef7df7348d"; }
