"""Constants of the reference's ``config.py`` that the hot path consumes (values cited per line)."""
CLIP_MODEL = "openai/clip-vit-large-patch14-336"      # config.py:6
CLIP_EMBED_DIM = 1024                                 # config.py:7
TINYVIT_MODEL = "tiny_vit_21m_512.dist_in22k_ft_in1k"  # config.py:9
DECAY_CONSTANT = 1492.7                               # config.py:49
LABEL_SMOOTHING_CONSTANT = 65                         # config.py:52
CURRENT_SAVE_PATH = "saved_models/WorldCLIP_head_landmarks.model"   # config.py:56
CLIP_PRETRAINED_HEAD = "saved_models/New_Base_smooth_avg_MT_Geo_SV.model"   # config.py:60
EMBED_BATCH_SIZE_PER_GPU = 512                        # config.py:63
NUM_GEOCELLS = 12647                                  # shipped geocell pickles
NUM_ATTENTION_HEADS = 16                              # config.py (hierarchical SuperGuessr, models/super_guessr.py:91-97)
