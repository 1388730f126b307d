extern "C" const char* vd_source_sha(void) { return "cd831a1a1c3fe141"; }
