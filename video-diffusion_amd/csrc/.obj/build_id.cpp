extern "C" const char* vd_source_sha(void) { return "071f328f42b4f42d"; }
