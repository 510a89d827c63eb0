"""Patches consistent_depth_amd/csrc/loss_sweep.hip IN PLACE so that every workgroup leaves five wall-clock stamps (kernel entry, after the
pair constants, loop entry, loop exit, after the epilogue) in the first 40 bytes of its pair's gradient (wrong results by construction;
never commit the patched file).  Usage (from the repo root):
    cp consistent_depth_amd/csrc/loss_sweep.hip /tmp/keep.hip && python tools/exp/sweep_times_patch.py
    tools/exp/build_variants.sh times consistent_depth_amd/csrc/loss_sweep.hip -fno-slp-vectorize
    cp /tmp/keep.hip consistent_depth_amd/csrc/loss_sweep.hip
then CD_AMD_LIB=tools/exp/variants/libcd_amd_times.so python tools/exp/sweep_times.py on the GPU box."""
import os
p = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'consistent_depth_amd', 'csrc', 'loss_sweep.hip')
s=open(p).read()
def rep(old,new,cnt=1):
    global s
    assert old and s.count(old)==cnt, (s.count(old), old[:60])
    s=s.replace(old,new)
rep('''    if (threadIdx.x == 0) { st.ovf_n = 0u; st.redo = 0; st.is_last = 0; }''','''    unsigned long long tstamp[6];
    tstamp[0] = wall_clock64();
    if (threadIdx.x == 0) { st.ovf_n = 0u; st.redo = 0; st.is_last = 0; }''')
rep('''    if (threadIdx.x < 2 * (int)(sizeof(PairCam) / sizeof(float)))      // (kept in the workspace: debugging, the tile kernels' format)''','''    tstamp[1] = wall_clock64();
    if (threadIdx.x < 2 * (int)(sizeof(PairCam) / sizeof(float)))      // (kept in the workspace: debugging, the tile kernels' format)''')
rep('''    if (!two) {
        // ONE pass per item''','''    tstamp[2] = wall_clock64();
    if (!two) {
        // ONE pass per item''')
rep('''    if (env.any(r.bad) && (threadIdx.x & (kWave - 1)) == 0) env.degenerate();''','''    tstamp[3] = wall_clock64();
    if (env.any(r.bad) && (threadIdx.x & (kWave - 1)) == 0) env.degenerate();''')
rep('''    if (st.is_last != 0) {
        double* ld''','''    tstamp[4] = wall_clock64();
    if (threadIdx.x == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(grad + (size_t)b * 2 * HW);
        for (int i = 0; i < 5; ++i) o[i] = tstamp[i];
    }
    if (st.is_last != 0) {
        double* ld''')
open(p,'w').write(s)
