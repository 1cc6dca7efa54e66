// The reference headers include "ceres/rotation.h" (src/lba_problem.h:26, src/po_problem.h:25) for the
// templated functors.  The functors are evaluated on the GPU here, so nothing from it is needed on
// the host; the header exists so that the include line compiles.
#ifndef SLSLAM_HOST_CERES_ROTATION_H_
#define SLSLAM_HOST_CERES_ROTATION_H_
#endif
