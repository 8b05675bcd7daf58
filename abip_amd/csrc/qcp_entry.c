/* qcp_entry.c -- the conic entry points under the REFERENCE'S OWN names (src/abip-qcp/include/abip.h:235-241, source/util.c:203-255):
 *     abip_int abip(const ABIPData *d, ABIPSolution *sol, ABIPInfo *info, ABIPCone *K);
 *     void     abip_set_default_settings(ABIPData *d);
 * for libabip_hip_qcp.so (Makefile): the same objects as libabip_hip.so with the LP entry of the same name renamed out of the way, exporting
 * exactly these symbols (exports_qcp.map), so that the reference's conic gateways (mex/abip_qcp_mex.c, mex/abip_ml_mex.c) link against it with no
 * source change beyond dropping their two MKL-pulling includes.  The structs are the reference's conic layouts (include/abip_qcp.h mirrors them). */
#include "../../include/abip_qcp.h"

__attribute__((visibility("default"))) qcp_int abip(const QCPData *d, QCPSolution *sol, QCPInfo *info, QCPCone *K) { return abip_qcp(d, sol, info, K); }
__attribute__((visibility("default"))) void abip_set_default_settings(QCPData *d) { abip_qcp_set_default_settings(d); }
