"""Patches consistent_depth_amd/csrc/loss_sweep.hip IN PLACE with the switches of round 5's timing experiments (results are wrong by
construction; never commit the patched file):  -DCD_SWEEP_EXP=n, bits: 1 no source pass, 2 no flush, 4 no staging by the sources, 8 one
item per pair.  Usage (from the repo root):
    cp consistent_depth_amd/csrc/loss_sweep.hip /tmp/keep.hip && python tools/exp/sweep_exp_patch.py
    for n in 1 2 3 8; do tools/exp/build_variants.sh exp$n consistent_depth_amd/csrc/loss_sweep.hip -fno-slp-vectorize -DCD_SWEEP_EXP=$n; done
    cp /tmp/keep.hip consistent_depth_amd/csrc/loss_sweep.hip
then tools/exp/r05_exp.sh on the GPU box.  The replacements assert that their anchors exist exactly once: a kernel that has moved on
makes the script fail instead of patching the wrong place."""
import os, sys
p = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'consistent_depth_amd', 'csrc', 'loss_sweep.hip')
s=open(p).read()
def rep(old,new,cnt=1):
    global s
    assert old and s.count(old)==cnt, (s.count(old), old[:60])
    s=s.replace(old,new)
rep('''            svc_flush<NQ>(v, slc, sl, me.fl_lo, me.fl_hi);
            __syncthreads();''','''#if !(CD_SWEEP_EXP & 2)
            svc_flush<NQ>(v, slc, sl, me.fl_lo, me.fl_hi);
#endif
            __syncthreads();''')
rep('''            if constexpr (FAST) {
                if (me.p >= 0) process_rows_fast''','''#if !(CD_SWEEP_EXP & 1)
            if constexpr (FAST) {
                if (me.p >= 0) process_rows_fast''')
rep('''            } else process_rows<MODE, REPROJ, PXT>(v, env, r, l, cur, me.p, 0, wk, nvk);
            __syncthreads();
            me = nx; wk = nwk; nvk = nnvk;
            nx = n2;''','''            } else process_rows<MODE, REPROJ, PXT>(v, env, r, l, cur, me.p, 0, wk, nvk);
#else
            r.pend_r += cur.fx[0] + cur.fy[0] + cur.m[0] + cur.fx[1] + cur.fy[1] + cur.m[1];
#endif
            __syncthreads();
            me = nx; wk = nwk; nvk = nnvk;
            nx = n2;''')
rep('''            if (SRC_STAGES) {      // the depth rows entering now were requested during the previous item; request the next ones
''','''            if (SRC_STAGES && !(CD_SWEEP_EXP & 4)) {      // the depth rows entering now were requested during the previous item; request the next ones
''')
rep('''    const int n_items = ph->n_items;
''','''#if (CD_SWEEP_EXP & 8)
    const int n_items = ph->n_items > 0 ? 1 : 0;
#else
    const int n_items = ph->n_items;
#endif
''')
s=s.replace('namespace cd {','#ifndef CD_SWEEP_EXP\n#define CD_SWEEP_EXP 0\n#endif\nnamespace cd {',1)
open(p,'w').write(s)
