from gym.utils import seeding
