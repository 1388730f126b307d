extern "C" const char* vd_source_sha(void) { return "ae0b9ab56542b183"; }
