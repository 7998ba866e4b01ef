"""One rank of tests/test_gpu_session.py::test_comm_two_ranks_over_mock_transport (run as a script, one process per rank):
    python tests/_comm_rank.py <rank> <world> <workdir>
Rank 0 owns the clips, every rank recognises its shard through the C-ABI session, rank 0 checks the gathered transcripts
against the Python surface on all clips and writes <workdir>/ok."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(rank, world, work):
    from danspeech_amd import _native
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_gpu_session as t
    rec, sd = t._engine(seed=17)
    ses, keep = t._session(rec, sd)
    idf = os.path.join(work, "id")
    if rank == 0:
        uid = _native.NativeComm.unique_id()
        with open(idf + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(idf + ".tmp", idf)
    else:
        for _ in range(6000):
            if os.path.exists(idf):
                break
            time.sleep(0.01)
        uid = open(idf, "rb").read()
    comm = _native.NativeComm(uid, rank, world, 0)
    lengths = [9000, 30000, 9000, 41000, 16000, 22222, 8000]
    for rnd, count in enumerate((len(lengths), 1)):                 # a full batch, then fewer clips than ranks
        clips = t._clips(lengths[:count], seed0=20 * rnd) if rank == 0 else None
        dev, n, pos, code, total = comm.scatter(clips)
        assert total == count and len(n) == len(pos) == (count - rank + world - 1) // world if count > rank else len(n) == 0
        assert all(n[i] >= n[i + 1] for i in range(len(n) - 1))
        if len(n):
            ses.enqueue_device(dev, n, code)
            text, _, _ = ses.collect(raw=True)
        else:
            text = np.zeros((0, 4096), dtype=np.uint8)
        out = comm.gather_text(text, pos, total)
        if rank == 0:
            assert out == rec.recognize_batch(clips), (out, rnd)
        else:
            assert out is None
    comm.close()
    ses.close()
    if rank == 0:
        open(os.path.join(work, "ok"), "w").write("ok")


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), sys.argv[3])
