// kicp_hsaco.S -- embeds the gfx950 code object of kicp_reg_launch.hip (build/kicp_reg.hsaco: the same translation unit compiled
// device-only) into libkicp_amd.so, so that kicp_aql.hpp can hand it to the HSA loader and dispatch the pass kernels with
// hand-written AQL packets.  Read-only data, 4 KiB aligned.
    .section .rodata.kicp_hsaco, "a", @progbits
    .balign 4096
    .globl kicp_hsaco_start
    .hidden kicp_hsaco_start
kicp_hsaco_start:
#ifdef KICP_DBG_BUILD
    .incbin "build_dbg/kicp_reg.hsaco"
#else
    .incbin "build/kicp_reg.hsaco"
#endif
    .globl kicp_hsaco_end
    .hidden kicp_hsaco_end
kicp_hsaco_end:
    .balign 8
    .section .note.GNU-stack, "", @progbits
