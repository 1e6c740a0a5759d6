/*****************************************************************
 * Planar two-link arm (both joints actuated, angles measured from the horizontal) -- an EXAMPLE plant file written in the plug-in form of
 * plancherb1/parallel-DDP (plants/dynamics_cart.cuh / dynamics_arm.cuh are the models for the form; nothing of them is reused):
 *
 *   initI(T *s_I), initT(T *s_T)                                            constant tables, handed back to the functions below as d_I / d_Tbody
 *   dynamics(T *s_qdd, T *s_x, T *s_u, T *d_I, T *d_Tbody, T *s_eePos = nullptr, int reps = 1, T *s_eeVel = nullptr)
 *   dynamicsGradient(T *s_dqdd, T *s_qdd, T *s_x, T *s_u, T *d_I, T *d_Tbody)          s_dqdd[col*NUM_POS + row], cols = q1 q2 qd1 qd2 u1 u2
 *
 * Build it into the library as plant 5 (csrc/ref_plugin.hpp):
 *   make -C parallel-ddp_amd user PLANT_FILE=examples/plants/dynamics_twolink.cuh COST_FILE=examples/plants/cost_twolink.cuh NUM_POS=2 CONTROL_SIZE=2 NUM_TIME_STEPS=64 USER_TAG=twolink
 *
 * M(q) qdd + c(q,qd) + g(q) + D qd = u with
 *   M = [a + 2 b cos q2, d + b cos q2; d + b cos q2, d],  a = I1 + I2 + m1 r1^2 + m2 (l1^2 + r2^2),  b = m2 l1 r2,  d = I2 + m2 r2^2
 *   c = b sin q2 [-(2 qd1 qd2 + qd2^2); qd1^2],   g = grav [(m1 r1 + m2 l1) cos q1 + m2 r2 cos(q1+q2); m2 r2 cos(q1+q2)]
 * The link parameters live in the table initI fills (slots below), so the d_I path of the plug-in interface is exercised.
 *****************************************************************/
#ifndef TWOLINK_GRAVITY
	#define TWOLINK_GRAVITY 9.81
#endif
// slots of the parameter table
#define TL_M1 0
#define TL_M2 1
#define TL_L1 2
#define TL_R1 3
#define TL_R2 4
#define TL_I1 5
#define TL_I2 6
#define TL_G 7
#define TL_D1 8
#define TL_D2 9

template <typename T> __host__ __device__ __forceinline__
void initI(T *s_I){
	s_I[TL_M1] = 1.2;    s_I[TL_M2] = 0.8;
	s_I[TL_L1] = 0.5;    s_I[TL_R1] = 0.25;    s_I[TL_R2] = 0.2;
	s_I[TL_I1] = 0.03;   s_I[TL_I2] = 0.015;
	s_I[TL_G]  = TWOLINK_GRAVITY;
	s_I[TL_D1] = 0.3;    s_I[TL_D2] = 0.2;     // viscous joint damping
}
template <typename T> __host__ __device__ __forceinline__ void initT(T *s_T){return;}     // no transforms in this plant

// the pieces both functions need: mass matrix entries, joint torques after bias, and 1/det
template <typename T> __host__ __device__ __forceinline__
void twolink_terms(T *t, T *s_xk, T *s_uk, T *d_I){
	T m1 = d_I[TL_M1];   T m2 = d_I[TL_M2];   T l1 = d_I[TL_L1];   T r1 = d_I[TL_R1];   T r2 = d_I[TL_R2];
	T a = d_I[TL_I1] + d_I[TL_I2] + m1*r1*r1 + m2*(l1*l1 + r2*r2);
	T b = m2*l1*r2;      T d = d_I[TL_I2] + m2*r2*r2;
	T c2 = cos(s_xk[1]); T s2 = sin(s_xk[1]);
	T c1 = cos(s_xk[0]); T c12 = cos(s_xk[0] + s_xk[1]);
	T qd1 = s_xk[2];     T qd2 = s_xk[3];
	T M11 = a + 2*b*c2;  T M12 = d + b*c2;    T M22 = d;
	T g2 = m2*r2*d_I[TL_G]*c12;               T g1 = (m1*r1 + m2*l1)*d_I[TL_G]*c1 + g2;
	T tau1 = s_uk[0] + b*s2*(2*qd1*qd2 + qd2*qd2) - g1 - d_I[TL_D1]*qd1;
	T tau2 = s_uk[1] - b*s2*qd1*qd1 - g2 - d_I[TL_D2]*qd2;
	t[0] = M11;  t[1] = M12;  t[2] = M22;  t[3] = tau1;  t[4] = tau2;  t[5] = 1/(M11*M22 - M12*M12);  t[6] = b;  t[7] = s2;  t[8] = c2;
}

template <typename T>
__host__ __device__ __forceinline__
void dynamics(T *s_qdd, T *s_x, T *s_u, T *d_I, T *d_Tbody, T *s_eePos = nullptr, int reps = 1, T *s_eeVel = nullptr){
	int start, delta; singleLoopVals(&start,&delta);
	for(int iter = start; iter < reps; iter += delta){
		T *s_xk = &s_x[STATE_SIZE*iter];   T *s_uk = &s_u[CONTROL_SIZE*iter];   T *s_qddk = &s_qdd[NUM_POS*iter];
		T t[9];  twolink_terms(t,s_xk,s_uk,d_I);
		s_qddk[0] = t[5] * (t[2]*t[3] - t[1]*t[4]);
		s_qddk[1] = t[5] * (t[0]*t[4] - t[1]*t[3]);
	}
	hd__syncthreads();
}

template <typename T>
__host__ __device__ __forceinline__
void dynamicsGradient(T *s_dqdd, T *s_qdd, T *s_x, T *s_u, T *d_I, T *d_Tbody){
	#ifdef __CUDA_ARCH__
		if (threadIdx.x != 0 || threadIdx.y != 0){return;}
	#endif
	if (s_qdd != nullptr){dynamics(s_qdd,s_x,s_u,d_I,d_Tbody);}
	T t[9];  twolink_terms(t,s_x,s_u,d_I);
	T M11 = t[0];  T M12 = t[1];  T M22 = t[2];  T tau1 = t[3];  T tau2 = t[4];  T idet = t[5];  T b = t[6];  T s2 = t[7];  T c2 = t[8];
	T qd1 = s_x[2];    T qd2 = s_x[3];
	T N1 = M22*tau1 - M12*tau2;        T N2 = M11*tau2 - M12*tau1;
	T m1 = d_I[TL_M1]; T m2 = d_I[TL_M2]; T l1 = d_I[TL_L1]; T r1 = d_I[TL_R1]; T r2 = d_I[TL_R2]; T grav = d_I[TL_G];
	T s1 = sin(s_x[0]);                T s12 = sin(s_x[0] + s_x[1]);
	// d/dq1: only gravity moves
	T dt1 = (m1*r1 + m2*l1)*grav*s1 + m2*r2*grav*s12;   T dt2 = m2*r2*grav*s12;
	s_dqdd[0] = idet * (M22*dt1 - M12*dt2);             s_dqdd[1] = idet * (M11*dt2 - M12*dt1);
	// d/dq2: mass matrix, Coriolis and gravity
	T dM11 = -2*b*s2;                  T dM12 = -b*s2;
	T ddet = dM11*M22 - 2*M12*dM12;
	dt1 = b*c2*(2*qd1*qd2 + qd2*qd2) + m2*r2*grav*s12;  dt2 = -b*c2*qd1*qd1 + m2*r2*grav*s12;
	T dN1 = M22*dt1 - dM12*tau2 - M12*dt2;              T dN2 = dM11*tau2 + M11*dt2 - dM12*tau1 - M12*dt1;
	s_dqdd[2] = idet * (dN1 - idet*N1*ddet);            s_dqdd[3] = idet * (dN2 - idet*N2*ddet);
	// d/dqd1, d/dqd2
	dt1 = 2*b*s2*qd2 - d_I[TL_D1];     dt2 = -2*b*s2*qd1;
	s_dqdd[4] = idet * (M22*dt1 - M12*dt2);             s_dqdd[5] = idet * (M11*dt2 - M12*dt1);
	dt1 = 2*b*s2*(qd1 + qd2);          dt2 = -d_I[TL_D2];
	s_dqdd[6] = idet * (M22*dt1 - M12*dt2);             s_dqdd[7] = idet * (M11*dt2 - M12*dt1);
	// d/du = M^-1
	s_dqdd[8] = idet * M22;            s_dqdd[9] = -idet * M12;
	s_dqdd[10] = -idet * M12;          s_dqdd[11] = idet * M11;
}
