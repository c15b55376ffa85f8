// Internal layout constants of the tile-VM pass descriptor (mirrored in tcmi/plan.py).
//
// desc[0]=MAGIC desc[1]=n desc[2]=T desc[3]=R desc[4]=LT desc[5]=nrounds desc[8..8+T)=tile bit
// positions (ascending physical bits).  Then per round a record of TCMI_RR_WORDS words
//   [0] nops  [1] op words  [2..8) reg phys masks  [8..18) thread phys masks
//   [18..24) reg LDS-read masks  [24..34) thread LDS-read masks   (exchange into this round)
//   [34..40) reg LDS-write masks [40..50) thread LDS-write masks  (exchange out of this round)
// followed by the round's ops:
//   G1   {1, j, slot}                 dense 1-qubit gate on register bit j
//   G2   {2, ja, jb, slot}            dense 2-qubit gate, matrix index = (bit ja << 1) | bit jb, ja < jb
//   DIAG {3, nA, nB, nC, A.., B.., C..}  phase polynomial in turns:
//        A = {mask, slot}      coef * (-1)^parity(thread_index & mask)
//        B = {j, mask, slot}   coef * (-1)^parity(thread_index & mask) * z_j(r)
//        C = {rmask, slot}     coef * (-1)^parity(r & rmask)
// slot = offset (in reals) into the per-batch table, or into the constant table if TCMI_CONST_FLAG.
#ifndef TCMI_VM_H
#define TCMI_VM_H

#include "../../include/tcmi.h"

#define TCMI_MAGIC 0x54434D31
#define TCMI_HDR_WORDS 24
#define TCMI_RR_WORDS 50
#define TCMI_OP_G1 1
#define TCMI_OP_G2 2
#define TCMI_OP_DIAG 3
#define TCMI_CONST_FLAG (1 << 30)
#define TCMI_BK_TRIG 1
#define TCMI_BK_COEF 2

#endif
