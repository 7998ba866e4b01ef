"""CPU restatement of the reference's offline long-form segmentation (TEST INFRASTRUCTURE ONLY).

Follows example_scripts/video_transcribe_simulation.py:68-143 line by line: a simulated stream of
1024-sample chunks, RMS energy per chunk, a phrase opens on the first chunk above the energy
threshold (two chunks of lead-in when available) and closes after more than
ceil(pause_threshold / chunk_seconds) quiet chunks; it is kept when it held more than
ceil(phrase_threshold / chunk_seconds) chunks besides that pause.

Pinning status: PINNED to the reference itself.  The script is a command-line example whose gate lives in its
``__main__`` block and cannot be imported as a function, so tools/gen_golden_segments.py EXECUTES the script
(runpy, in the build container) on seeded WAV files with a recording stand-in for ``Recognizer`` and stores the
sample ranges it hands to ``recognize()``: tests/golden/g9_segments.json.  tests/test_oracle_segmentation.py holds
this restatement to those ranges; tests/test_gpu_recognizer.py holds ``dsmi_segment`` to them directly.
"""
import numpy as np


def segment(audio, energy_threshold=600, step=1024, pause_threshold=0.55, phrase_threshold=0.2, sampling_rate=16000):
    """-> (list of (start_index, end_index) sample ranges, float64 energies per chunk)."""
    audio = np.asarray(audio, dtype=np.float64)
    iterator = 0
    pause_buffer_count = np.ceil(pause_threshold / (step / sampling_rate))       # :77
    phrase_buffer_count = np.ceil(phrase_threshold / (step / sampling_rate))     # :81
    is_speaking = False
    frames_counter = 0
    pause_count = 0
    start_index = 0
    out, energies = [], []
    while (iterator + step) < len(audio):                                        # :94
        temp_data = audio[iterator:iterator + step]
        energy = np.sqrt((temp_data * temp_data).sum() / (1. * len(temp_data)))  # :100
        energies.append(energy)
        if energy > energy_threshold and not is_speaking:                        # :103-113
            is_speaking = True
            start_index = iterator - 2 * step
            if start_index < 0:
                start_index = iterator
        iterator += step                                                         # :116
        if is_speaking:                                                          # :118-125
            frames_counter += 1
            if energy > energy_threshold:
                pause_count = 0
            else:
                pause_count += 1
        if pause_count > pause_buffer_count and is_speaking:                     # :128-143
            if (frames_counter - pause_count) > phrase_buffer_count:
                out.append((start_index, iterator))
            is_speaking = False
            frames_counter = 0
            pause_count = 0
    return out, np.array(energies, dtype=np.float64)
