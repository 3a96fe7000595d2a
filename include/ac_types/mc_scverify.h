// mc_scverify.h -- stand-alone definitions of the Catapult SCVerify macros.
// The ac_dsp class templates declare their top function as CCS_BLOCK(run)
// (reference include/ac_dsp/ac_fir_const_coeffs.h:321); outside Catapult the
// macro is the identity (cf. reference tests/mc_scverify.h).
#ifndef AC_DSP_AMD_MC_SCVERIFY_H
#define AC_DSP_AMD_MC_SCVERIFY_H
#if defined(CCS_SCVERIFY) || defined(CCS_SYSC)
#error ac_dsp_amd ships only the stand-alone mc_scverify.h; use the Catapult header for SCVerify flows
#endif
#ifndef CCS_BLOCK
#define CCS_BLOCK(a) a
#endif
#ifndef CCS_MAIN
#define CCS_MAIN(a, b) int main(a, b)
#endif
#ifndef CCS_RETURN
#define CCS_RETURN(a) return (a)
#endif
#ifndef CCS_DESIGN
#define CCS_DESIGN(a) a
#endif
#endif
