// Named landing registers for loads that stay in flight across barriers (conv1d_bsplit.hip, conv1d_bsplit2.hip).
//
// hipcc does not know that the destination of an inline-asm load is invalid until the matching s_waitcnt, so a value that is
// compiler-visible while its load is in flight can be copied or spilled too early (seen: copies at control-flow joins, wrong
// results only when the memory system is loaded).  The kernels therefore land such loads in PHYSICAL registers named in the asm
// text, at the top of the 256-register budget their launch bounds give them, and read them back -- after the s_waitcnt -- with the
// v_cndmask that zeroes padding lanes anyway.  The X-macro tables below list (element index, register number) pairs;
// tools/check_inflight_regs.py verifies on the ISA that no compiler-generated instruction touches the named range.
#pragma once

// two sets of 24 registers: v208..v231, v232..v255
#define FAC_XREGS24_A(M) M(0, 208) M(1, 209) M(2, 210) M(3, 211) M(4, 212) M(5, 213) M(6, 214) M(7, 215) M(8, 216) M(9, 217) M(10, 218) \
  M(11, 219) M(12, 220) M(13, 221) M(14, 222) M(15, 223) M(16, 224) M(17, 225) M(18, 226) M(19, 227) M(20, 228) M(21, 229) M(22, 230) M(23, 231)
#define FAC_XREGS24_B(M) M(0, 232) M(1, 233) M(2, 234) M(3, 235) M(4, 236) M(5, 237) M(6, 238) M(7, 239) M(8, 240) M(9, 241) M(10, 242) \
  M(11, 243) M(12, 244) M(13, 245) M(14, 246) M(15, 247) M(16, 248) M(17, 249) M(18, 250) M(19, 251) M(20, 252) M(21, 253) M(22, 254) M(23, 255)
// two sets of 40 registers: v176..v215, v216..v255
#define FAC_XREGS40_A(M) M(0, 176) M(1, 177) M(2, 178) M(3, 179) M(4, 180) M(5, 181) M(6, 182) M(7, 183) M(8, 184) M(9, 185) M(10, 186) \
  M(11, 187) M(12, 188) M(13, 189) M(14, 190) M(15, 191) M(16, 192) M(17, 193) M(18, 194) M(19, 195) M(20, 196) M(21, 197) M(22, 198) M(23, 199) \
  M(24, 200) M(25, 201) M(26, 202) M(27, 203) M(28, 204) M(29, 205) M(30, 206) M(31, 207) M(32, 208) M(33, 209) M(34, 210) M(35, 211) M(36, 212) \
  M(37, 213) M(38, 214) M(39, 215)
#define FAC_XREGS40_B(M) M(0, 216) M(1, 217) M(2, 218) M(3, 219) M(4, 220) M(5, 221) M(6, 222) M(7, 223) M(8, 224) M(9, 225) M(10, 226) \
  M(11, 227) M(12, 228) M(13, 229) M(14, 230) M(15, 231) M(16, 232) M(17, 233) M(18, 234) M(19, 235) M(20, 236) M(21, 237) M(22, 238) M(23, 239) \
  M(24, 240) M(25, 241) M(26, 242) M(27, 243) M(28, 244) M(29, 245) M(30, 246) M(31, 247) M(32, 248) M(33, 249) M(34, 250) M(35, 251) M(36, 252) \
  M(37, 253) M(38, 254) M(39, 255)
