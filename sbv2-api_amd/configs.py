"""Model shapes used by tests and bench.py (host side; the library itself reads the shapes from the weight container).

The same dictionaries exist in oracle/sbv2_oracle.py (the oracle is self-contained test infrastructure and is not imported from
here); tests/test_host_cpu.py checks that the two copies agree."""

#: ku-nlp/deberta-v2-large-japanese-char-wwm as exported by convert_deberta.py:11,25 — output is
#: hidden_states[-3] of 25 = the state after layer 22 (convert_deberta.py:34), so only 22 layers run.
#: conv_kernel_size / conv_act = DebertaV2Encoder.conv, the ConvLayer after layer 0 (recalled for the ku-nlp checkpoint, see the oracle).
DEBERTA_FULL = dict(
    vocab_size=22012, hidden=1024, layers=22, heads=16, intermediate=4096,
    position_buckets=256, max_relative_positions=512, ln_eps=1e-7, conv_kernel_size=3, conv_act="gelu",
)
DEBERTA_TINY = dict(
    vocab_size=96, hidden=64, layers=2, heads=4, intermediate=128,
    position_buckets=8, max_relative_positions=32, ln_eps=1e-7, conv_kernel_size=0, conv_act="gelu",
)
DEBERTA_TINY_CONV = dict(DEBERTA_TINY, conv_kernel_size=3, conv_act="gelu")

#: Style-Bert-VITS2 JP-Extra defaults (configs/config_jp_extra.json upstream); n_vocab = 112 symbols
#: (crates/sbv2_core/src/norm.rs:57-96), tones 12, languages 3.
VITS_FULL = dict(
    n_vocab=112, n_tones=12, n_langs=3, n_speakers=1,
    hidden=192, inter=192, filter=768, heads=2, enc_layers=6, enc_kernel=3, window=4,
    gin=512, style_dim=256, bert_dim=1024, cond_layer_idx=2,
    flow_n=4, flow_layers=6, flow_kernel=5,
    dp_filter=256, dp_kernel=3,
    sdp_kernel=3, sdp_flows=4, sdp_bins=10, sdp_tail=5.0, sdp_dds_layers=3,
    up_rates=[8, 8, 2, 2, 2], up_kernels=[16, 16, 8, 2, 2], up_initial=512,
    res_kernels=[3, 7, 11], res_dilations=[[1, 3, 5], [1, 3, 5], [1, 3, 5]],
)
VITS_TINY = dict(
    n_vocab=112, n_tones=12, n_langs=3, n_speakers=2,
    hidden=32, inter=32, filter=64, heads=2, enc_layers=3, enc_kernel=3, window=4,
    gin=16, style_dim=8, bert_dim=64, cond_layer_idx=2,
    flow_n=2, flow_layers=3, flow_kernel=5,
    dp_filter=48, dp_kernel=3,
    sdp_kernel=3, sdp_flows=4, sdp_bins=10, sdp_tail=5.0, sdp_dds_layers=3,
    up_rates=[4, 2, 2], up_kernels=[8, 4, 2], up_initial=64,
    res_kernels=[3, 7], res_dilations=[[1, 3, 5], [1, 3, 5]],
)

SAMPLE_RATE = 44100  # crates/sbv2_core/src/tts_util.rs:164-169
