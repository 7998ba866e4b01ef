"""Write a model package (state dict + configuration + labels + audio configuration) as the flat binary file the
non-Python example host reads (examples/host_recognize.c):

    "DSMIW001"
    int32  conv_layers, rnn_type (0 gru / 1 lstm / 2 rnn), rnn_hidden_size, rnn_layers, bidirectional, context,
           n_labels, sample_rate, window (0 hamming / 1 hann / 2 blackman / 3 bartlett), normalize
    double window_size, window_stride
    n_labels x { int32 n_bytes, UTF-8 bytes }
    int32  n_tensors
    n_tensors x { int32 name_bytes, name, int32 ndim, int64 shape[ndim], float32 data[prod(shape)] }

All little-endian.  Tensor names are the reference's state-dict keys (what dsmi_model_load_tensor takes).

    python tools/export_weights.py model.pth out.dsmiw        # a package saved by DeepSpeech.serialize / the reference
"""
import struct
import sys

import numpy as np

RNN = {"gru": 0, "lstm": 1, "rnn": 2}
WIN = {"hamming": 0, "hann": 1, "blackman": 2, "bartlett": 3}


def write_pack(path, state_dict, cfg, labels, audio_conf):
    with open(path, "wb") as f:
        f.write(b"DSMIW001")
        f.write(struct.pack("<10i", cfg["conv_layers"], RNN[cfg["rnn_type"]], cfg["rnn_hidden_size"], cfg["rnn_layers"],
                            int(cfg.get("bidirectional", True)), int(cfg.get("context", 20)), len(labels),
                            int(audio_conf.get("sampling_rate", 16000)), WIN[audio_conf.get("window", "hamming")],
                            int(audio_conf.get("normalize", True))))
        f.write(struct.pack("<2d", float(audio_conf.get("window_size", 0.02)), float(audio_conf.get("window_stride", 0.01))))
        for lab in labels:
            b = lab.encode("utf-8")
            f.write(struct.pack("<i", len(b)) + b)
        items = [(k, np.ascontiguousarray(np.asarray(v, dtype=np.float32))) for k, v in state_dict.items()
                 if not k.endswith("num_batches_tracked")]
        f.write(struct.pack("<i", len(items)))
        for name, a in items:
            nb = name.encode("utf-8")
            f.write(struct.pack("<i", len(nb)) + nb + struct.pack("<i", a.ndim) + struct.pack("<%dq" % a.ndim, *a.shape))
            f.write(a.tobytes())


def main(argv):
    if len(argv) != 3:
        print(__doc__)
        return 2
    import torch
    pkg = torch.load(argv[1], map_location="cpu", weights_only=True)
    sd = {k: v.numpy() for k, v in pkg["state_dict"].items()}
    cfg = dict(conv_layers=pkg["conv_layers"], rnn_type=pkg["rnn_type"], rnn_hidden_size=pkg["rnn_hidden_size"], rnn_layers=pkg["rnn_layers"],
               bidirectional=pkg.get("bidirectional", True), context=pkg.get("context", 20))
    write_pack(argv[2], sd, cfg, list(pkg["labels"]), pkg["audio_conf"])
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
