/*****************************************************************
 * Cost file for the two-link example (dynamics_twolink.cuh) in the plug-in form of plancherb1/parallel-DDP's CURRENT joint-space cost
 * (plants/cost_arm.cuh:130,158: the five weights arrive as trailing arguments at run time) -- with a Hessian that is NOT diagonal:
 *
 *   running:  1/2 [ Q1 sum e_i^2 + Q2 sum qd_i^2 + QC (e1 - e2)^2 + R sum (u_i + KD qd_i)^2 + RC (u1 - u2)^2 ],   e = q - q_goal
 *   final:    1/2 [ QF1 sum e_i^2 + QF2 sum qd_i^2 + QFC (e1 - e2)^2 ]
 *
 * QC couples the two joints (off-diagonal state block), RC the two torques (off-diagonal control block), KD ties a torque to its joint's velocity
 * (the state-control cross blocks) -- costGrad writes every block, both triangles, column-major with leading dimension ld_H.
 *****************************************************************/
#if EE_COST
	#error "the two-link example has no end effector cost -- compile with EE_COST turned off."
#endif
#ifndef _Q1
	#define _Q1 0.5
	#define _Q2 0.01
	#define _R  0.001
	#define _QF1 500.0
	#define _QF2 50.0
#endif
#ifndef TWOLINK_QC
	#define TWOLINK_QC 0.2
	#define TWOLINK_QFC 100.0
	#define TWOLINK_RC 0.0005
	#define TWOLINK_KD 0.5
#endif

template <typename T>
__host__ __device__ __forceinline__
T costFunc(T *xk, T *uk, T *xgk, int k, T Q1 = _Q1, T Q2 = _Q2, T R = _R, T QF1 = _QF1, T QF2 = _QF2){
	T cost = 0.0;
	T e1 = xk[0]-xgk[0];   T e2 = xk[1]-xgk[1];
	if (k == NUM_TIME_STEPS - 1){
		cost += QF1*(e1*e1 + e2*e2) + QF2*(xk[2]*xk[2] + xk[3]*xk[3]) + static_cast<T>(TWOLINK_QFC)*(e1-e2)*(e1-e2);
	}
	else{
		T w1 = uk[0] + static_cast<T>(TWOLINK_KD)*xk[2];   T w2 = uk[1] + static_cast<T>(TWOLINK_KD)*xk[3];
		cost += Q1*(e1*e1 + e2*e2) + Q2*(xk[2]*xk[2] + xk[3]*xk[3]) + static_cast<T>(TWOLINK_QC)*(e1-e2)*(e1-e2);
		cost += R*(w1*w1 + w2*w2) + static_cast<T>(TWOLINK_RC)*(uk[0]-uk[1])*(uk[0]-uk[1]);
	}
	return static_cast<T>(0.5)*cost;
}

template <typename T>
__host__ __device__ __forceinline__
void costGrad(T *Hk, T *gk, T *xk, T *uk, T *xgk, int k, int ld_H, T Q1 = _Q1, T Q2 = _Q2, T R = _R, T QF1 = _QF1, T QF2 = _QF2){
	T e1 = xk[0]-xgk[0];   T e2 = xk[1]-xgk[1];
	if (k == NUM_TIME_STEPS - 1){
		T qc = static_cast<T>(TWOLINK_QFC);
		#pragma unroll
		for (int i=0; i<STATE_SIZE; i++){
			#pragma unroll
			for (int j=0; j<STATE_SIZE; j++){Hk[i*ld_H + j] = (i != j) ? static_cast<T>(0) : (i < NUM_POS ? QF1 + qc : QF2);}
		}
		Hk[0*ld_H + 1] = -qc;   Hk[1*ld_H + 0] = -qc;
		gk[0] = QF1*e1 + qc*(e1-e2);   gk[1] = QF1*e2 - qc*(e1-e2);   gk[2] = QF2*xk[2];   gk[3] = QF2*xk[3];
		#pragma unroll
		for (int i=0; i<CONTROL_SIZE; i++){gk[i+STATE_SIZE] = 0;}
	}
	else{
		T qc = static_cast<T>(TWOLINK_QC);   T rc = static_cast<T>(TWOLINK_RC);   T kd = static_cast<T>(TWOLINK_KD);
		T w1 = uk[0] + kd*xk[2];   T w2 = uk[1] + kd*xk[3];
		#pragma unroll
		for (int i=0; i<STATE_SIZE+CONTROL_SIZE; i++){
			#pragma unroll
			for (int j=0; j<STATE_SIZE+CONTROL_SIZE; j++){
				Hk[i*ld_H + j] = (i != j) ? static_cast<T>(0) : (i < NUM_POS ? Q1 + qc : (i < STATE_SIZE ? Q2 + R*kd*kd : R + rc));
			}
		}
		Hk[0*ld_H + 1] = -qc;          Hk[1*ld_H + 0] = -qc;            // joint coupling
		Hk[4*ld_H + 5] = -rc;          Hk[5*ld_H + 4] = -rc;            // torque coupling
		Hk[2*ld_H + 4] = R*kd;         Hk[4*ld_H + 2] = R*kd;           // qd1 <-> u1
		Hk[3*ld_H + 5] = R*kd;         Hk[5*ld_H + 3] = R*kd;           // qd2 <-> u2
		gk[0] = Q1*e1 + qc*(e1-e2);    gk[1] = Q1*e2 - qc*(e1-e2);
		gk[2] = Q2*xk[2] + R*kd*w1;    gk[3] = Q2*xk[3] + R*kd*w2;
		gk[4] = R*w1 + rc*(uk[0]-uk[1]);   gk[5] = R*w2 - rc*(uk[0]-uk[1]);
	}
}
