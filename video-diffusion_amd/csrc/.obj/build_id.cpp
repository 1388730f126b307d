extern "C" const char* vd_source_sha(void) { return "87fce257449bf7c4"; }
