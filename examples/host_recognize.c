/* A host without Python: recognise PCM WAV files through the C ABI alone (include/dsmi.h).
 *
 *   gcc -std=c99 -O2 -Iinclude examples/host_recognize.c -o host_recognize -Ldanspeech_amd/lib -ldsmi -Wl,-rpath,$PWD/danspeech_amd/lib
 *   ./host_recognize model.dsmiw a.wav b.wav ...              greedy
 *   ./host_recognize model.dsmiw --lm lm.klm --alpha 1.3 --beta 0.2 --beam 64 a.wav ...
 *
 * model.dsmiw: tools/export_weights.py.  The WAV files (16-bit PCM, all mono or all stereo) are handed over as their raw
 * frames: the channel fold and every later step run on the GPU.  One transcript per line, in argument order -- what
 * `Recognizer(model=...).recognize(load_audio(path))` prints for each file (reference example_scripts/execute_recognize.py). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dsmi.h"

static void die(const char* what, const char* why) {
    fprintf(stderr, "host_recognize: %s: %s\n", what, why ? why : "");
    exit(1);
}

static void rd(FILE* f, void* p, size_t n) {
    if (fread(p, 1, n, f) != n) die("model file", "truncated");
}

/* 16-bit PCM WAV -> malloc'ed raw frames; returns the frame count */
static int64_t read_wav(const char* path, int* channels, int* rate, void** frames) {
    FILE* f = fopen(path, "rb");
    unsigned char h[12], ck[8];
    int have_fmt = 0;
    if (!f) die(path, "cannot open");
    if (fread(h, 1, 12, f) != 12 || memcmp(h, "RIFF", 4) || memcmp(h + 8, "WAVE", 4)) die(path, "not a RIFF/WAVE file");
    while (fread(ck, 1, 8, f) == 8) {
        const uint32_t size = (uint32_t)ck[4] | (uint32_t)ck[5] << 8 | (uint32_t)ck[6] << 16 | (uint32_t)ck[7] << 24;
        if (!memcmp(ck, "fmt ", 4)) {
            unsigned char fm[16];
            if (size < 16 || fread(fm, 1, 16, f) != 16) die(path, "bad fmt chunk");
            if ((fm[0] | fm[1] << 8) != 1 || (fm[14] | fm[15] << 8) != 16) die(path, "only 16-bit PCM is handled here");
            *channels = fm[2] | fm[3] << 8;
            *rate = (int)((uint32_t)fm[4] | (uint32_t)fm[5] << 8 | (uint32_t)fm[6] << 16 | (uint32_t)fm[7] << 24);
            fseek(f, (long)(size - 16 + (size & 1)), SEEK_CUR);
            have_fmt = 1;
        } else if (!memcmp(ck, "data", 4)) {
            if (!have_fmt || *channels < 1 || *channels > 2) die(path, "unsupported channel layout");
            *frames = malloc(size ? size : 1);
            if (!*frames || fread(*frames, 1, size, f) != size) die(path, "truncated data chunk");
            fclose(f);
            return (int64_t)(size / (2u * (unsigned)*channels));
        } else {
            fseek(f, (long)(size + (size & 1)), SEEK_CUR);
        }
    }
    die(path, "no data chunk");
    return 0;
}

int main(int argc, char** argv) {
    const char* lm = NULL;
    double alpha = 1.3, beta = 0.2;
    int beam = 0, first = 2, i;
    if (argc < 3) die("usage", "host_recognize model.dsmiw [--lm path --alpha a --beta b --beam n] file.wav ...");
    while (first + 1 < argc && argv[first][0] == '-' && argv[first][1] == '-') {
        if (!strcmp(argv[first], "--lm")) lm = argv[first + 1];
        else if (!strcmp(argv[first], "--alpha")) alpha = atof(argv[first + 1]);
        else if (!strcmp(argv[first], "--beta")) beta = atof(argv[first + 1]);
        else if (!strcmp(argv[first], "--beam")) beam = atoi(argv[first + 1]);
        else die("unknown option", argv[first]);
        first += 2;
    }
    if (lm && !beam) beam = 64;                       /* the engine's default width, reference DanSpeechRecognizer.py:29 */

    /* ---- the model package */
    FILE* f = fopen(argv[1], "rb");
    char magic[8];
    int32_t hd[10], n_tensors;
    double win[2];
    if (!f) die(argv[1], "cannot open");
    rd(f, magic, 8);
    if (memcmp(magic, "DSMIW001", 8)) die(argv[1], "not a DSMIW001 file");
    rd(f, hd, sizeof hd);
    rd(f, win, sizeof win);
    dsmi_model_desc md;
    md.conv_layers = hd[0]; md.rnn_type = hd[1]; md.rnn_hidden_size = hd[2]; md.rnn_layers = hd[3]; md.bidirectional = hd[4];
    md.context = hd[5]; md.n_labels = hd[6]; md.sample_rate = hd[7]; md.window_size = win[0];
    dsmi_frontend_desc fd;
    fd.sample_rate = hd[7]; fd.window_size = win[0]; fd.window_stride = win[1]; fd.window = hd[8]; fd.normalize = hd[9];
    fd.pad_mode = DSMI_PAD_REFLECT;
    char** labels = (char**)calloc((size_t)hd[6], sizeof(char*));
    int blank = 0;
    for (i = 0; i < hd[6]; ++i) {
        int32_t nb;
        rd(f, &nb, 4);
        labels[i] = (char*)calloc((size_t)nb + 1, 1);
        rd(f, labels[i], (size_t)nb);
        if (!strcmp(labels[i], "_")) blank = i;         /* labels.index('_'), reference DanSpeechRecognizer.py:92 */
    }
    dsmi_model* model = NULL;
    if (dsmi_model_create(&md, 0, &model)) die("dsmi_model_create", dsmi_last_error(NULL));
    rd(f, &n_tensors, 4);
    for (i = 0; i < n_tensors; ++i) {
        int32_t nb, ndim, k;
        int64_t shape[4], count = 1;
        char name[256];
        rd(f, &nb, 4);
        if (nb < 1 || nb > 255) die(argv[1], "bad tensor name");
        rd(f, name, (size_t)nb);
        name[nb] = 0;
        rd(f, &ndim, 4);
        if (ndim < 0 || ndim > 4) die(name, "bad rank");
        rd(f, shape, sizeof(int64_t) * (size_t)ndim);
        for (k = 0; k < ndim; ++k) count *= shape[k];
        float* data = (float*)malloc(sizeof(float) * (size_t)(count ? count : 1));
        rd(f, data, sizeof(float) * (size_t)count);
        if (dsmi_model_load_tensor(model, name, data, shape, ndim)) die(name, dsmi_last_error(model));
        free(data);
    }
    fclose(f);
    if (dsmi_model_finalize(model)) die("dsmi_model_finalize", dsmi_last_error(model));

    dsmi_frontend* fe = NULL;
    dsmi_decoder* dec = NULL;
    dsmi_session* ses = NULL;
    if (dsmi_frontend_create(&fd, 0, &fe)) die("dsmi_frontend_create", dsmi_frontend_last_error(NULL));
    if (dsmi_decoder_create(0, (const char* const*)labels, hd[6], blank, &dec)) die("dsmi_decoder_create", dsmi_decoder_last_error(NULL));
    if (lm && dsmi_decoder_set_lm(dec, lm, alpha, beta)) die("dsmi_decoder_set_lm", dsmi_decoder_last_error(dec));
    if (dsmi_session_create(fe, model, dec, &ses)) die("dsmi_session_create", dsmi_session_last_error(NULL));

    /* ---- the recordings */
    const int B = argc - first;
    const void** clips = (const void**)calloc((size_t)B, sizeof(void*));
    int64_t* n = (int64_t*)calloc((size_t)B, sizeof(int64_t));
    int channels0 = 0;
    for (i = 0; i < B; ++i) {
        int ch = 0, rate = 0;
        void* frames = NULL;
        n[i] = read_wav(argv[first + i], &ch, &rate, &frames);
        clips[i] = frames;
        if (rate != hd[7]) die(argv[first + i], "sample rate differs from the model's");
        if (i && ch != channels0) die(argv[first + i], "mono and stereo files in one batch");
        channels0 = ch;
    }
    const int stride = 4096;
    char* text = (char*)calloc((size_t)B, (size_t)stride);
    const int rc = dsmi_recognize_batch(ses, clips, n, DSMI_PCM_I16 | (channels0 == 2 ? DSMI_PCM_STEREO : 0), B, beam, 40, 1.0, text, stride,
                                        NULL, NULL);
    if (rc < 0) die("dsmi_recognize_batch", dsmi_session_last_error(ses));
    for (i = 0; i < B; ++i) printf("%s\n", text + (size_t)i * stride);

    dsmi_session_destroy(ses);
    dsmi_decoder_destroy(dec);
    dsmi_frontend_destroy(fe);
    dsmi_model_destroy(model);
    return 0;
}
