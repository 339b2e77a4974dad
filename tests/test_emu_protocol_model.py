"""A small model checker for round 6's stream-emulation protocol (CPU, no GPU): the words LAP_H_VERDICT and LAP_H_EMU between the solver's kernel,
the sparse emulation's kernel on the shared emulation stream and the final kernel / patch step.  Every interleaving of the atomic steps of
two consecutive launch chains is explored; the properties: each frame is committed by exactly ONE writer, nobody waits for ever, and a provisional
commit is only made while the emulation's kernel is resident.  The model restates the device code it cites and is test infrastructure only; what it is
good for: the first version of the protocol ended the emulation's kernel with a plain store of "finished" -- that variant is explored too and MUST
produce the double commit the two-context soak found after ~2,000 repetitions (profiles/r06_prov_soak.log)."""
import itertools

import pytest


def tagged_word(seq, v):                      # assoc_common.h: tagged_word
    return ((seq & 0x3FFFFFFF) << 2) | v


def tagged_value(word, seq, newer):           # assoc_common.h: tagged_value
    t, s = (word & 0xFFFFFFFF) >> 2, seq & 0x3FFFFFFF
    if t == s:
        return word & 3
    return newer if ((t - s) & 0x3FFFFFFF) < 0x20000000 else 0


class Violation(Exception):
    pass


def explore(seqs, truth, plain_store):
    """seqs: the two chains' sequence numbers; truth[i]: what chain i's solver finds (1 certified, 2 tie it cannot commit, 3 tie of the provisional shape).
    State: (emu word, verdict word, main pc, emu pc, per-chain commit counts, per-chain flags).  Main stream program per chain: ROWSCAN, SOLVE, FINAL, PATCH;
    emulation stream program per chain: START, RUN, FINISH (lap_kernels.hip: mk_sparse_stream_kernel)."""
    MAIN = [(c, op) for c in range(2) for op in ("rowscan", "solve", "final", "patch")] + [(None, "end")]
    EMU = [(c, op) for c in range(2) for op in ("start", "run", "finish")] + [(None, "end")]
    init = (0, 0, 0, 0, (0, 0), ((0, 0, 0), (0, 0, 0)))     # flags per chain: (verdict published by the solver, emulation's kernel skipped, provisional)
    seen, stack, finals = set(), [init], 0
    while stack:
        st = stack.pop()
        if st in seen:
            continue
        seen.add(st)
        emu, ver, mpc, epc, commits, flags = st
        succ = []
        # ---- main stream ----
        c, op = MAIN[mpc]
        if op == "rowscan":                                 # lap_rowscan_kernel re-arms the verdict word (lap_kernels.hip:45)
            succ.append((emu, 0, mpc + 1, epc, commits, flags))
        elif op == "solve":                                 # lap_solve_run + lap_try_provisional + the tagged verdict (lap_kernels.hip:270, 682)
            v = truth[c]
            if v == 3 and tagged_value(emu, seqs[c], 0) != 1:
                v = 2                                       # the emulation's kernel is not resident: no provisional commit
            cm = list(commits)
            if v in (1, 3):
                cm[c] += 1
            fl = list(flags); fl[c] = (v, flags[c][1], 1 if v == 3 else 0)
            succ.append((emu, tagged_word(seqs[c], v), mpc + 1, epc, tuple(cm), tuple(fl)))
        elif op == "final":                                 # munkres_kernel<*, false>, stream mode (assoc_kernels.hip: the wait / claim loop)
            if flags[c][0] != 2:
                succ.append((emu, ver, mpc + 1, epc, commits, flags))
            else:
                stv = tagged_value(emu, seqs[c], 2)
                if stv == 0:                                # not started: take the frame away from the emulation's kernel (CAS; single step = atomic)
                    cm = list(commits); cm[c] += 1
                    succ.append((tagged_word(seqs[c], 3), ver, mpc + 1, epc, tuple(cm), flags))
                elif stv == 2:                              # finished (or a later chain's word): bookkeeping if it committed the frame (LAP_H_DONE), else the
                    cm = list(commits)                      # dense emulation decides it here -- e.g. a run that gave up on a verdict word it took for a newer chain's
                    if cm[c] == 0:
                        cm[c] = 1
                    succ.append((emu, ver, mpc + 1, epc, tuple(cm), flags))
                elif stv == 3:
                    raise Violation(f"final kernel of chain {c} finds its own claim")
                # stv == 1: started -> wait (no successor from this process)
        elif op == "patch":                                 # munkres_kernel<false, true>: waits for "finished" of a provisionally committed frame
            if not flags[c][2] or tagged_value(emu, seqs[c], 2) == 2:
                succ.append((emu, ver, mpc + 1, epc, commits, flags))
        # ---- emulation stream (in order: chain 1's kernel behind chain 0's; a kernel is behind its chain's row scan) ----
        c, op = EMU[epc]
        if op != "end" and mpc > MAIN.index((c, "rowscan")):
            if op == "start":
                if tagged_value(emu, seqs[c], 3) == 3:      # claimed (or a later chain is running): nothing to do
                    fl = list(flags); fl[c] = (flags[c][0], 1, flags[c][2])
                    succ.append((emu, ver, mpc, epc + 3, commits, tuple(fl)))
                else:                                       # CAS old -> started (a failed CAS re-reads: the same step again, so one atomic step here)
                    succ.append((tagged_word(seqs[c], 1), ver, mpc, epc + 1, commits, flags))
            elif op == "run":                               # mk_sparse_run<SPEC>: ends by sp_wait_verdict (mk_sparse_body.h:81)
                v = tagged_value(ver, seqs[c], 1)
                if v != 0:                                  # 0: pending -> keep polling
                    cm = list(commits)
                    if v == 2:
                        cm[c] += 1                          # decides AND commits the frame
                    succ.append((emu, ver, mpc, epc + 1, tuple(cm), flags))
            elif op == "finish":
                if plain_store or emu == tagged_word(seqs[c], 1):
                    succ.append((tagged_word(seqs[c], 2), ver, mpc, epc + 1, commits, flags))
                else:
                    succ.append((emu, ver, mpc, epc + 1, commits, flags))
        if not succ:
            if MAIN[mpc][1] == "end" and EMU[epc][1] == "end":
                finals += 1
                if commits != (1, 1):
                    raise Violation(f"commits per chain {commits} (truth {truth}, seqs {seqs})")
                continue
            raise Violation(f"deadlock at main {MAIN[mpc]} emu {EMU[epc]} (truth {truth})")
        for s2 in succ:
            if max(s2[4]) > 1:
                raise Violation(f"frame committed twice: {s2[4]} (truth {truth}, seqs {seqs})")
            stack.append(s2)
    return len(seen), finals


# (the host counter skips 0: launch_assoc.  Chain 0x3FFFFFFF reads the RE-ARMED verdict word -- an untagged 0 -- as a newer chain's: its emulation gives
# up at once and the final kernel's dense emulation decides that frame, once in 2^30 launches; the model covers it)
SEQS = [(5, 6), (0x3FFFFFFF, 1), (0x3FFFFFFE, 0x3FFFFFFF)]


@pytest.mark.parametrize("seqs", SEQS)
def test_every_interleaving_commits_each_frame_exactly_once(seqs):
    total = 0
    for truth in itertools.product((1, 2, 3), repeat=2):
        states, finals = explore(seqs, truth, plain_store=False)
        assert finals > 0
        total += states
    assert total > 100


def test_plain_store_of_finished_is_the_double_commit_the_soak_found():
    """the first version: a late kernel of a certified / provisional chain overwrites the next chain's claim, that chain's own kernel starts after all"""
    with pytest.raises(Violation, match="committed twice"):
        for truth in itertools.product((1, 2, 3), repeat=2):
            explore((5, 6), truth, plain_store=True)
