// placeholder until the pose-graph kernels land (next commit)
#include "../../include/slslam_hip.h"
extern "C" int slslam_po_solve(const slslam_po_graph*, const slslam_solver_options*, slslam_summary*,
                               slslam_iteration*, int, int*) { return SLSLAM_ERR_UNSUPPORTED; }
