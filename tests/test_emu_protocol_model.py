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


def explore(seqs, truth, plain_store, tagged_rearm=True):
    """seqs: the consecutive chains' sequence numbers; truth[i]: what chain i's solver finds (1 certified, 2 tie it cannot commit, 3 tie of the provisional shape).
    State: (emu word, verdict word, main pc, emu pc, per-chain commit counts, per-chain flags).  Main stream program per chain: ROWSCAN, SOLVE, FINAL, PATCH;
    emulation stream program per chain: START, RUN, FINISH (lap_kernels.hip: mk_sparse_stream_kernel)."""
    n = len(seqs)
    MAIN = [(c, op) for c in range(n) for op in ("rowscan", "solve", "final", "patch")] + [(None, "end")]
    EMU = [(c, op) for c in range(n) for op in ("start", "run", "finish")] + [(None, "end")]
    # the words as the previous chain left them (a fresh context: zero); flags per chain: (verdict published by the solver, emulation's kernel skipped, provisional)
    prev = (seqs[0] - 1) & 0x3FFFFFFF
    init = (tagged_word(prev, 2) if prev else 0, tagged_word(prev, 1) if prev else 0, 0, 0, (0,) * n, ((0, 0, 0),) * n)
    seen, stack, finals, writers = set(), [init], 0, set()

    def commit(have, who):                                  # who: 1 the solver's workgroup, 2 the emulation's kernel, 3 the final kernel after its claim, 4 the final kernel behind a run that ended without committing
        if have:
            raise Violation(f"frame committed twice: by {have} and by {who} (truth {truth}, seqs {seqs})")
        return who
    while stack:
        st = stack.pop()
        if st in seen:
            continue
        seen.add(st)
        emu, ver, mpc, epc, commits, flags = st
        succ = []
        # ---- main stream ----
        c, op = MAIN[mpc]
        if op == "rowscan":                                 # lap_rowscan_kernel re-arms the verdict word as this chain's "pending" (lap_kernels.hip:46)
            succ.append((emu, tagged_word(seqs[c], 0) if tagged_rearm else 0, mpc + 1, epc, commits, flags))
        elif op == "solve":                                 # lap_solve_run + lap_try_provisional + the tagged verdict (lap_kernels.hip:270, 682)
            v = truth[c]
            if v == 3 and tagged_value(emu, seqs[c], 0) != 1:
                v = 2                                       # the emulation's kernel is not resident: no provisional commit
            cm = list(commits)
            if v in (1, 3):
                cm[c] = commit(cm[c], 1)
            fl = list(flags); fl[c] = (v, flags[c][1], 1 if v == 3 else 0)
            succ.append((emu, tagged_word(seqs[c], v), mpc + 1, epc, tuple(cm), tuple(fl)))
        elif op == "final":                                 # munkres_kernel<*, false>, stream mode (assoc_kernels.hip: the wait / claim loop)
            if flags[c][0] != 2:
                succ.append((emu, ver, mpc + 1, epc, commits, flags))
            else:
                stv = tagged_value(emu, seqs[c], 2)
                if stv == 0:                                # not started: take the frame away from the emulation's kernel (CAS; single step = atomic)
                    cm = list(commits); cm[c] = commit(cm[c], 3)
                    succ.append((tagged_word(seqs[c], 3), ver, mpc + 1, epc, tuple(cm), flags))
                elif stv == 2:                              # finished (or a later chain's word): bookkeeping if it committed the frame (LAP_H_DONE), else the
                    cm = list(commits)                      # dense emulation decides it here -- e.g. a run that gave up on a verdict word it took for a newer chain's
                    if cm[c] == 0:
                        cm[c] = commit(cm[c], 4)
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
                        cm[c] = commit(cm[c], 2)            # decides AND commits the frame
                    succ.append((emu, ver, mpc, epc + 1, tuple(cm), flags))
            elif op == "finish":
                if plain_store or emu == tagged_word(seqs[c], 1):
                    succ.append((tagged_word(seqs[c], 2), ver, mpc, epc + 1, commits, flags))
                else:
                    succ.append((emu, ver, mpc, epc + 1, commits, flags))
        if not succ:
            if MAIN[mpc][1] == "end" and EMU[epc][1] == "end":
                finals += 1
                if 0 in commits:
                    raise Violation(f"a frame was never committed: {commits} (truth {truth}, seqs {seqs})")
                writers.add(commits)
                continue
            raise Violation(f"deadlock at main {MAIN[mpc]} emu {EMU[epc]} (truth {truth})")
        stack.extend(succ)
    return len(seen), finals, writers


# (the host counter skips 0: launch_assoc)
SEQS = [(5, 6), (0x3FFFFFFF, 1), (0x3FFFFFFE, 0x3FFFFFFF)]


@pytest.mark.parametrize("seqs", SEQS)
def test_every_interleaving_commits_each_frame_exactly_once(seqs):
    total = 0
    for truth in itertools.product((1, 2, 3), repeat=2):
        states, finals, writers = explore(seqs, truth, plain_store=False)
        assert finals > 0
        assert all(4 not in w for w in writers), (truth, writers)       # a run that started is never talked out of deciding its frame
        total += states
    assert total > 100


def test_plain_store_of_finished_is_the_double_commit_the_soak_found():
    """the first version: a late kernel of a certified / provisional chain overwrites the next chain's claim, that chain's own kernel starts after all"""
    with pytest.raises(Violation, match="committed twice"):
        for truth in itertools.product((1, 2, 3), repeat=2):
            explore((5, 6), truth, plain_store=True)


def test_untagged_rearm_costs_chain_0x3fffffff_its_sparse_emulation():
    """what this model found in the first version of the re-arm (the row scan stored an untagged 0): chain 0x3FFFFFFF reads that word as a NEWER chain's
    verdict, its emulation gives up at once and the final kernel's dense emulation has to decide the frame -- still exactly one writer, but the slow
    tier, once in 2^30 launches.  The row scan now stores the chain's own "pending": wherever the emulation's kernel got to run, it decides."""
    seqs, truth = (0x3FFFFFFF, 1), (2, 1)
    _, _, untagged = explore(seqs, truth, plain_store=False, tagged_rearm=False)
    _, _, tagged = explore(seqs, truth, plain_store=False, tagged_rearm=True)
    assert {w[0] for w in untagged} == {2, 3, 4}                         # 4: the run read the re-armed 0 as "a newer chain's verdict" and gave up
    assert {w[0] for w in tagged} == {2, 3}                              # the emulation's kernel when it started in time, the final kernel's claim otherwise


def test_three_chains():
    """a kernel of the emulation stream can be two chains late (both certified: nobody waits for it)"""
    for truth in itertools.product((1, 2, 3), repeat=3):
        _, finals, writers = explore((7, 8, 9), truth, plain_store=False)
        assert finals > 0 and all(4 not in w for w in writers), (truth, writers)
    with pytest.raises(Violation, match="committed twice"):
        for truth in itertools.product((1, 2, 3), repeat=3):
            explore((7, 8, 9), truth, plain_store=True)
