"""Configuration surface of the SELD training path (drop-in for the reference's ``config.py``).

Every attribute the reference defines (config.py:7-97) is present with the same name, type and
default, and ``Config()`` creates the output / checkpoint directories and the derived dataset
paths exactly like config.py:99-118 (main.py and trainer.py rely on CHECKPOINT_PATH existing).
Attributes below the "MI355X additions" line do not exist upstream; they default to values that
keep upstream semantics on a CPU and switch the GPU fast paths on when a ROCm device is used.
"""
from pathlib import Path

_SR = 24000


class Config:
    # ---- locations (relative to this file, like the reference) ---------------------------
    BASE_PATH = Path(__file__).parent
    AUDIO_PATH = BASE_PATH / "foa_dev"
    METADATA_PATH = BASE_PATH / "metadata_dev"
    OUTPUT_PATH = BASE_PATH / "outputs"
    CHECKPOINT_PATH = BASE_PATH / "checkpoints"

    # ---- data selection -----------------------------------------------------------------
    USE_FULL_DATASET = True
    TRAIN_AUDIO_FILE = "fold3_room21_mix001.wav"
    TRAIN_META_FILE = "fold3_room21_mix001.csv"
    TEST_AUDIO_FILE = "fold4_room23_mix001.wav"
    TEST_META_FILE = "fold4_room23_mix001.csv"

    STARSS22_CLASSES = dict(enumerate((
        "Female speech, woman speaking", "Male speech, man speaking", "Clapping", "Telephone",
        "Laughter", "Domestic sounds", "Walk, footsteps", "Door, open or close", "Music",
        "Musical instrument", "Water tap, faucet", "Bell", "Knock", "Background")))

    # ---- model ---------------------------------------------------------------------------
    MODEL_TYPE = "resnet_conformer"      # 'cnn' | 'crnn' | 'conformer' | 'resnet_conformer'
    NUM_CLASSES = 14
    N_CHANNELS = 4

    CRNN_CNN_CHANNELS = [64, 128, 256, 512]
    CRNN_RNN_HIDDEN = 256
    CRNN_RNN_LAYERS = 2
    CRNN_DROPOUT = 0.3

    CONF_D_MODEL = 256
    CONF_N_HEADS = 4
    CONF_N_LAYERS = 2
    CONF_KERNEL_SIZE = 31
    CONF_DROPOUT = 0.3

    RESNET_CONF_D_MODEL = 512
    RESNET_CONF_N_HEADS = 8
    RESNET_CONF_N_LAYERS = 4
    RESNET_DROPOUT = 0.3

    # ---- optimisation --------------------------------------------------------------------
    NUM_EPOCHS = 30
    BATCH_SIZE = 16
    LEARNING_RATE = 1e-3
    LR_DECAY_FACTOR = 0.5
    LR_DECAY_PATIENCE = 5
    WEIGHT_DECAY = 1e-4

    LOSS_TYPE = "mse"                    # 'mse' | 'ce'
    W_CLASS = 1.0
    W_AIUR = 1.0
    W_CL = 1.0

    PATIENCE = 20
    MIN_DELTA = 1e-4

    SAVE_EVERY_N_EPOCHS = 5
    KEEP_LAST_N_CHECKPOINTS = 3

    # ---- signal processing: 40 ms frames, 20 ms hop at 24 kHz ------------------------------
    SPECTROGRAM_N_FFT = int(0.04 * _SR)          # 960
    SPECTROGRAM_HOP_LENGTH = int(0.02 * _SR)     # 480
    N_MELS = 64
    SR = _SR

    # ---- windowing of the concatenated timeline: 5 s windows, 1 s hop ----------------------
    WINDOW_LENGTH = int(5 * _SR)                 # 120000 samples = 250 frames
    HOP_LENGTH = int(1 * _SR)                    # 24000 samples  = 50 frames

    # ---- DOA grid ------------------------------------------------------------------------
    I = None
    J = None
    GRID_CELL_DEGREES = 10

    # ==== MI355X additions (not in the reference) ==========================================
    AMP_DTYPE = "bf16"          # autocast dtype on ROCm devices: 'bf16' or 'fp32'
    CHANNELS_LAST = True        # NHWC conv blocks (MIOpen/hipBLASLt MFMA path)
    FUSED_LOSS = True           # seld_softmax_mse instead of softmax + mse_loss + autograd
    FUSED_CONV_TAIL = True      # BatchNorm -> ReLU -> MaxPool of the CNN blocks in two HBM passes (csrc/convtail.hip)
    CONV_DGRAD_AS_FORWARD = True  # encoder 3x3 convs: data gradient as a forward conv with transposed, flipped weights
    FUSED_DWCONV = "auto"       # channels-last Conformer conv module with the HIP depthwise Conv1d (csrc/dwconv.hip):
                                # fewer GPU microseconds but more host work; "auto" = always under GRAPH_STEP (no host work
                                # per iteration), else modules with d_model >= 512 (eager: +3 % on the ResNet50-Conformer,
                                # -7 % on the host-bound d_model-256 Conformer)
    CONV1X1_AS_GEMM = True      # ResNet bottlenecks: the 1x1 / stride-1 convolutions as GEMMs on the channels-last rows
                                # (hipBLASLt forward / data gradient, split-K weight gradient) instead of MIOpen's
    FUSED_QKV = True            # Conformer self-attention: the q / k / v projections of the same input as ONE GEMM on packed
                                # parameters (seld_pack.py), forward and backward
    FUSED_LAYERNORM = True      # head: LayerNorm -> ReLU in one kernel, activations stay bf16 (csrc/layernorm.hip)
    OVERLAP_WEIGHT_GRADS = True  # CRNN: weight gradients of the head and of GRU layer 1 on a side HIP stream, under the
                                # backward recurrences that occupy 16 of the 256 CUs (seld_overlap.py)
    FUSED_GRU = True            # persistent BiGRU kernel instead of MIOpen's per-step GEMMs
    DEVICE_FEED = True          # train from device-resident features / compact labels (no 290 MB/step H2D)
    MASTER_WEIGHTS = True       # bf16 runs: the conv / Linear / GRU weight matrices live in the model as bf16 working
                                # copies of fp32 masters owned by the optimiser (same values autocast would cast to
                                # every iteration, without ~40-340 cast kernels per iteration; bf16 gradient all-reduce)
    FUSED_ADAM_KERNEL = True    # master-weight mode, captured steps: the whole Adam update incl. the bf16 gradient read and the
                                # bf16 working-copy write as one multi-tensor launch (csrc/adam.hip) instead of cast + the
                                # framework's fused Adam + cast
    GRAPH_STEP = True           # one optimiser iteration = one replayed HIP graph per input shape (seld_graph.py): the eager
                                # loop spends ~3.5 ms of host time enqueueing ~300 launches per CRNN iteration; data
                                # parallel: one graph per backward stage, gradient buckets all-reduced in between
                                # (OVERLAP_ALLREDUCE below), then the update graph
    OVERLAP_ALLREDUCE = True    # data parallel + GRAPH_STEP: the backward pass is cut at the models' seld_cut.boundary points and
                                # captured as one graph per stage; the gradients a stage completes are all-reduced (RCCL,
                                # asynchronously) while the next stages run.  False: one graph, then one blocking exchange
    ALLREDUCE_CUT_LEVELS = 2    # 1: cut the backward pass only between the recurrent / attention part and the encoder; 2: also
                                # before the encoder's last block (one more bucket under way, ~50 us more per iteration:
                                # tools/bench_stages.py measured 3.62 ms as one graph, 3.67 / 3.72 / 3.77 ms with 0 / 1 / 2 cuts)
    GRAD_REDUCE_DTYPE = "param"  # wire dtype of that exchange: "param" = the bf16 working-weight gradients are summed in bf16
                                # (half the xGMI bytes), "fp32" = cast into fp32 buffers first (what an autocast port of the
                                # reference would reduce; the buffers double as the fp32 masters' gradients)
    TUNED_GEMMS = True          # apply tuned/gemm_gfx950.csv: hipBLASLt / rocBLAS kernel selections for the models' GEMM
                                # shapes, timed offline on an MI355X (seld_tuned.py; ignored on any other library stack)
    DDP_BUCKET_MB = 8           # RCCL all-reduce bucket size: with bf16 working weights the CRNN's gradients are 22 MB, so
                                # 8 MB gives three buckets -- the head's (ready first) reduces under the GRU / conv backward
    SYNC_BATCHNORM = False      # per-rank BN statistics by default
    SEED = None                 # the reference never seeds; set an int for reproducible runs
    FEATURE_SET = "logmel"      # 'logmel' (the reference) | 'logmel_iv' (FOA: + 3 intensity-vector channels) |
                                # 'logmel_gcc' (MIC array: + C(C-1)/2 GCC-PHAT channels); the model's n_channels
                                # follows the dataset (4 -> 7, 8 -> 36)
    FEATURE_CACHE_DIR = None    # directory for the per-recording COMPACT features (fp32 log-mel [T,C,64] + uint16 label mask
                                # [T,648], 2.3 KB per frame): a later SELDDataset construction uploads them instead of
                                # decoding and transforming the recording again
    THREE_TERM_LOSS = False     # total = W_CLASS * class + W_AIUR * AIUR + W_CL * CL on probabilities, as
                                # smrl_seld_gaussian.py:1058-1072 (BASELINE configs[4]: with GAUSSIAN_AUGMENT and the
                                # ResNet50-Conformer); False = the modular loss.py, class term only (loss.py:158-166)
    GAUSSIAN_AUGMENT = False    # smrl_seld_gaussian.py:397-534 label augmentation (+-2 sigma box per source)
    GAUSSIAN_SIGMA_AZIMUTH = 5.0
    GAUSSIAN_SIGMA_ELEVATION = 5.0

    def __init__(self):
        for folder in (self.OUTPUT_PATH, self.CHECKPOINT_PATH):
            folder.mkdir(exist_ok=True, parents=True)
        audio, meta = self.AUDIO_PATH, self.METADATA_PATH
        self.TRAIN_AUDIO_PATH = audio / "dev-train-sony" / self.TRAIN_AUDIO_FILE
        self.TRAIN_META_PATH = meta / "dev-train-sony" / self.TRAIN_META_FILE
        self.TEST_AUDIO_PATH = audio / "dev-test-sony" / self.TEST_AUDIO_FILE
        self.TEST_META_PATH = meta / "dev-test-sony" / self.TEST_META_FILE
        for site in ("SONY", "TAU"):
            for split in ("TRAIN", "TEST"):
                sub = f"dev-{split.lower()}-{site.lower()}"
                setattr(self, f"{site}_{split}_DIR", audio / sub)
                setattr(self, f"{site}_{split}_META_DIR", meta / sub)
