extern "C" const char* vd_source_sha(void) { return "fe6896cd3f4388c6"; }
